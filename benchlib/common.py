"""What every workload of bench.py shares: the peaks rooflines are priced against, the GPU's clocks as sysfs shows them, the
potentials of BASELINE's configurations, SURVEY 8(d)'s algorithmic bytes."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
PROF_EVERY = 7         # bracket every 7th force batch with hipEvents inside the timed region (14 samples per 100 steps)


def gpu_state(index=0):
    """Clocks and power cap of the GPU as sysfs shows them right now (a 10 % spread of an MFMA-bound kernel
    between boxes should be explained by a number).  Best effort: every field is optional."""
    import glob
    out = {}
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
    if not cards:
        return None
    dev = os.path.dirname(cards[min(index, len(cards) - 1)])

    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
        txt = read(os.path.join(dev, name))
        if txt:
            levels = [l.strip() for l in txt.splitlines()]
            cur = [l for l in levels if l.endswith("*")]
            out[name[7:] + "_now"] = cur[0].rstrip(" *").split(":")[-1].strip() if cur else None
            out[name[7:] + "_max"] = levels[-1].rstrip(" *").split(":")[-1].strip()
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for key, fname, scale in (("power_cap_W", "power1_cap", 1e-6), ("power_cap_max_W", "power1_cap_max", 1e-6),
                                  ("power_now_W", "power1_average", 1e-6), ("power_now_W", "power1_input", 1e-6),
                                  ("temp_edge_C", "temp1_input", 1e-3), ("sclk_hwmon_MHz", "freq1_input", 1e-6)):
            txt = read(os.path.join(hw, fname))
            if txt and key not in out:
                try:
                    out[key] = round(float(txt) * scale, 1)
                except ValueError:
                    pass
    out["perf_level"] = read(os.path.join(dev, "power_dpm_force_performance_level"))
    return out or None


def make_potential(htf, workload):
    if workload == "lj":
        return htf.Potential.lj()
    if workload == "wca":
        return htf.Potential.wca(1.0)
    from hoomd_tf_amd.initializers import mlp_params
    if workload == "mlp-train":
        # C5b = online force matching (example 06, FORCE_MODE::hoomd2tf): the reference LJ force
        # drives the MD; the pair-MLP is the model being trained, it does not push particles
        make_potential.layer = htf.PairMLP(32, 64, 64, 0.0, 3.0, activation="tanh", seed=3)
        return htf.Potential.lj()
    # "mlp": the default precision of PairMLP, fp32 operands as hi + lo in fp16 (DESIGN 3.3a''); the other three by name
    prec = {"mlp-bf16": "bf16", "mlp-split": "split", "mlp-fp32": "fp32"}.get(workload, "split16")
    return htf.Potential.pair_mlp(mlp_params(seed=3), 0.0, 3.0, activation="tanh", precision=prec)


def algorithmic_bytes(N, NN, n_list_entries, n_tot, s4=16):
    """per-launch algorithmic bytes of each kernel; s4 = bytes of a HOOMD Scalar4 (16 fp32, 32 fp64).
    The pair-vector tensor is fp32 either way."""
    eval_b = N * NN * 16 + N * s4
    build_b = N * 8 + n_list_entries * 4 + n_tot * s4 + N * NN * 16
    integ_b = N * s4 * 5  # pos r/w, vel r/w, force r
    return eval_b, build_b, integ_b
