"""The HOOMD-side shim (integration/hoomd_shim/) compiled against a FAKE of the HOOMD-blue 2.x headers
(integration/hoomd_stub/): HOOMD-blue is not in the image, so this file is a LINT -- syntax, types, overload
resolution, the pybind11 signatures, and that the module links against libhtf_amd.so and imports with
every method the reference exports (htf/TensorflowCompute.cc:422-486).  tests/test_gpu_shim.py RUNS the shim
(computeForces, period, batches, virial pitch, reference forces, training) against the same fake on a GPU."""
import os
import shutil
import subprocess
import sys
import sysconfig

import pytest

from helpers import ROOT

from helpers import SHIM, STUB, build_shim, shim_flags as _flags  # noqa: E402

# htf/TensorflowCompute.cc:431-481, one .def each
REFERENCE_EXPORTS = ["setMappedNlist", "getPositionsBuffer", "getNlistBuffer", "getForcesBuffer", "getBoxBuffer",
                     "getVirialBuffer", "getPositionsArray", "getNlistArray", "getForcesArray", "getBoxArray",
                     "getVirialArray", "isDoublePrecision", "getVirialPitch", "hook", "addReferenceForce"]


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("precision", ["double", "single"])
def test_shim_compiles_against_stub_headers(tmp_path, precision):
    extra = ["-DSINGLE_PRECISION"] if precision == "single" else []
    for src in ("TensorflowComputeAMD.cc", "module.cc"):
        r = subprocess.run(_flags(extra) + ["-c", os.path.join(SHIM, src), "-o", str(tmp_path / (src + ".o"))],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_shim_module_links_and_exports_the_reference_surface(tmp_path, htf):
    build_shim(tmp_path, htf._lib.LIB_PATH)
    code = ("import sys, torch; sys.path.insert(0, %r); import hoomd_tf_amd; sys.path.insert(0, %r); import _hoomd_stub; import _htf_amd as m; "
            "c = m.TensorflowComputeAMD; print(' '.join(n for n in dir(c) if not n.startswith('_'))); "
            "print(m.FORCE_MODE.tf2hoomd, m.FORCE_MODE.hoomd2tf, m.HalfStepHook.__name__); "
            "print(c.__init__.__doc__)" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    names = r.stdout.splitlines()[0].split()
    for n in REFERENCE_EXPORTS + ["setPotential", "setTraining", "getBatchCapacity"]:
        assert n in names, "shim does not export %s" % n
    assert "FORCE_MODE.tf2hoomd FORCE_MODE.hoomd2tf HalfStepHook" in r.stdout
    # the constructor takes the reference's eight arguments (TensorflowCompute.h:82-89)
    sig = r.stdout.split("__init__", 1)[1]
    for piece in ("SystemDefinition", "NeighborList", "FORCE_MODE"):
        assert piece in sig, sig
    assert sig.count("SupportsInt") + sig.count("int") >= 3 and "arg7" in sig
    # and it constructs against the stub objects: htf_create runs (host side), the getters answer
    code = ("import sys, torch; sys.path.insert(0, %r); import hoomd_tf_amd; sys.path.insert(0, %r); import _hoomd_stub as h; "
            "import _htf_amd as m; ok = torch.cuda.is_available(); "
            "c = m.TensorflowComputeAMD(object(), h.SystemDefinition(), h.NeighborList(), 3.0, 16, m.FORCE_MODE.tf2hoomd, 1, 0) "
            "if ok else None; print('constructed' if ok else 'no-gpu', (c.isDoublePrecision(), c.getBatchCapacity(), "
            "c.getNlistBuffer() != 0, type(c.hook()).__name__) if ok else '')" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and ("no-gpu" in r.stdout or "constructed (True, 1, True, 'HalfStepHook')" in r.stdout), r.stdout + r.stderr[-2000:]
