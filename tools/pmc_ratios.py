#!/usr/bin/env python3
"""Unit-busy fractions of the pair-MLP kernels from a committed PMC summary (profiles/rNN_bench_mlp_pmc.json):
matrix pipe = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1 024 SIMDs), vector unit = 4 x SQ_ACTIVE_INST_VALU over the
same denominator (the counter is in quad-cycles), share of a wave's life spent waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.
usage: python tools/pmc_ratios.py [profiles/r04_bench_mlp_pmc.json]"""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_bench_mlp_pmc.json"
d = json.load(open(path))
for section, kernels in d.items():
    if not isinstance(kernels, dict):
        continue
    print(section)
    for name, c in kernels.items():
        g = c.get("GRBM_GUI_ACTIVE", {}).get("avg")
        if not g:
            continue
        simd_cycles = g / 8.0 * 1024.0
        row = ["  %-58s" % name[:58], "%.2f ms at 2.2 GHz" % (g / 8.0 / 2.2e6)]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row.append("matrix pipe %.0f %%" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / simd_cycles))
        if "SQ_ACTIVE_INST_VALU" in c:
            row.append("vector unit %.0f %%" % (100.0 * 4.0 * c["SQ_ACTIVE_INST_VALU"]["avg"] / simd_cycles))
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            row.append("waiting %.0f %% of wave time" % (100.0 * c["SQ_WAIT_ANY"]["avg"] / c["SQ_WAVE_CYCLES"]["avg"]))
        print("  ".join(row))
