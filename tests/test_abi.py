"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/htf_amd.h declares, and its host-side argument validation maps to the
exception types the reference raises.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

from helpers import ROOT


def _declared_symbols(header="htf_amd.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"HTF_API[^;]*?\b(htfs?_\w+)\s*\(", hdr)))


def test_header_symbols_exported(htf):
    names = _declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(htf._lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libhtf_amd.so does not export %s" % n
    # and the python binding covers exactly the declared surface
    assert sorted(htf._lib.PROTOTYPES) == names
    assert raw.htf_abi_version() == 1
    standin = _declared_symbols("htf_standin.h")
    assert len(standin) >= 8 and sorted(htf._lib.STANDIN_PROTOTYPES) == standin
    for n in standin:
        assert hasattr(raw, n), "libhtf_amd.so does not export %s" % n


def test_header_is_plain_c():
    """The boundary must compile as C (no torch / C++ types in the signatures)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "htf_amd.h"\n#include "htf_standin.h"\nint main(void){htf_config c; (void)c; return HTF_OK;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src,
                               "-o", os.path.join(d, "t")])


def test_potential_validation(htf):
    with pytest.raises(ValueError):
        htf.Potential.wca(-1.0)
    with pytest.raises(ValueError):
        htf.Potential.rinv_poly([1.0], [0])
    with pytest.raises(ValueError):
        htf.Potential.rinv_poly([1.0] * 9, [1] * 9)
    with pytest.raises(ValueError):
        htf.Potential(99)
    p = htf.Potential.lj()
    assert p.handle


def test_device_tensor_required(htf):
    import torch
    with pytest.raises(ValueError):
        htf.ops.eval_forces(htf.Potential.lj(), torch.zeros(4, 8, 4))


def test_missing_library_fails_loudly(htf, tmp_path, monkeypatch):
    """No silent fallback: without libhtf_amd.so the binding refuses to import."""
    import importlib.util
    src = os.path.join(os.path.dirname(htf._lib.__file__), "_lib.py")
    dst = tmp_path / "_lib_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_lib_copy", str(dst))
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError):
        spec.loader.exec_module(mod)
