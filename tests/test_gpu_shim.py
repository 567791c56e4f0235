"""The HOOMD-side shim RUN on a GPU against a FAKE HOOMD (integration/hoomd_stub/: device-backed GlobalArray /
ArrayHandle, a ParticleData and a NeighborList the test points at its own device arrays; no integrator, no cell
list -- HOOMD-blue itself is not in the image).  tests/shim_driver.py constructs TensorflowComputeAMD from Python
the way hoomd/htf/tensorflowcompute.py:136-164 constructs TensorflowComputeGPU, drives computeForces for ten MD
steps per configuration (period, batch_size, virial + its pitch, MaxParticleNumberChange, hoomd2tf with the net
force and with addReferenceForce + setTraining through the half-step hook, error surfacing), and compares what it
leaves in "HOOMD's" m_force / m_virial with the htf.Context path bit for bit.
Reference body: htf/TensorflowCompute.cc:129-216, :250-301."""
import os
import shutil
import subprocess
import sys

import pytest

from helpers import ROOT, build_shim

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("precision", ["double", "single"])
def test_shim_runs_against_fake_hoomd(tmp_path, htf, cuda, precision):
    build_shim(tmp_path, htf._lib.LIB_PATH, single=(precision == "single"))
    r = subprocess.run([sys.executable, "-u", "-X", "faulthandler", os.path.join(ROOT, "tests", "shim_driver.py"), str(tmp_path), precision],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    for piece in ("OK forces virial=True batch_size=0 period=1", "OK forces virial=True batch_size=300 period=1",
                  "OK forces virial=False batch_size=0 period=3", "OK reallocate", "OK training n_ref=0", "OK training n_ref=2",
                  "OK errors"):
        assert piece in r.stdout, r.stdout
