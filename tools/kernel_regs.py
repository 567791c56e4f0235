#!/usr/bin/env python3
"""VGPR / SGPR-spill / scratch figures of the kernels in a libhtf_amd.so whose demangled name contains every given substring.
usage: tools/kernel_regs.py [--lib path] substr [substr ...]"""
import pathlib
import subprocess
import sys
import tempfile

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
import test_codeobj as t  # noqa: E402

args = sys.argv[1:]
if args and args[0] == "--lib":
    t.LIB = args[1]
    args = args[2:]
m = t._kernel_metadata(pathlib.Path(tempfile.mkdtemp()))
names = sorted(m)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
for n, d in zip(names, dem):
    short = d.split("(")[0]
    if all(a in short for a in args):
        v = m[n]
        print("%-70s vgpr %3d  sgpr_spill %3d  scratch %4d  lds %6d" % (short[-70:], v["vgpr_count"], v["sgpr_spill_count"],
                                                                       v["private_segment_fixed_size"], v["group_segment_fixed_size"]))
