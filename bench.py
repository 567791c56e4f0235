#!/usr/bin/env python3
"""Headline benchmark: MD steps/s + achieved HBM GB/s, 131 072 particles, NN = 128.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without WORLD_SIZE in the environment: this process starts N rank processes itself
        (fresh children, before anything touches the GPU) and relays rank 0's JSON line
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the whole particle batch, exactly what
HOOMD's run loop does around TensorflowCompute::computeForces (TensorflowCompute.cc:
129-216): neighbor-list distance check (rebuild when tripped), pair-vector build,
force/energy evaluation written into the HOOMD force array, and the integrator update
that moves the particles so the next step sees new input.  All inputs are resident in
HBM before the timed region starts.

Workload (SURVEY 8(d) "C3-LJ", BASELINE.json metric): fcc 4*32^3 = 131 072 particles,
rho = 0.8442, Gaussian jitter 0.05 a (seed 3), r_cut = 3.0, r_buff = 0.4, NN = 128,
LJModel, fp32, dt = 0.005, Maxwell velocities kT = 1.0.
N > 1, --scaling strong (default): THAT box decomposed into N slabs along x (BASELINE metric:
"131k particles ... at 1/2/4/8 GPUs"); --scaling weak: every rank owns one such block of an
N-block box (config 5: 8 x 131 072 = 1.05 M particles).  `value` is always the MD steps/s of the
GLOBAL system; `particle_steps_per_s` = global particles x value.

Prints ONE JSON line (rank 0).  roofline.achieved = ALGORITHMIC bytes per launch /
average launch duration measured with hipEvents on the launch stream inside the timed
region (htf_profile_*); see DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL's cross-process buffer sharing fails without this
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
PROF_EVERY = 7         # bracket every 7th force batch with hipEvents inside the timed region (see main; 14 samples per 100 steps)
from benchlib.common import HBM_PEAK_GBS, PROF_EVERY  # noqa: E402,F401
from benchlib.ddself import run_dd_self  # noqa: E402
from benchlib.eds import run_eds  # noqa: E402
from benchlib.generic import run_generic_lj  # noqa: E402
from benchlib.md import run_md  # noqa: E402
from benchlib.small import run_ref_lj256, run_small  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="lj", choices=["lj", "wca", "mlp", "mlp-fp32", "mlp-split", "mlp-bf16", "mlp-train", "eds", "ref-lj256", "c1", "ex01", "generic-lj", "dd-self"])
    ap.add_argument("--grid", default=None, help="rank grid PXxPYx1: --gpus N > 1: how the box is cut (default: slabs along x; e.g. 4x2x1 "
                                                 "for 8 ranks); dd-self: the grid whose one brick this GPU runs (default 8x1x1)")
    ap.add_argument("--transport", default="all", help="dd-self: local | peer | native | all")
    ap.add_argument("--replan-every", type=int, default=0,
                    help="BrickDomain(replan_every=k): only every k-th neighbor-list rebuild migrates and re-plans (a ghost layer (k - 1) r_buff "
                         "thicker).  0 = the default: 1 for --workload dd-self, 2 for --gpus N > 1 (there a re-plan is a grouped exchange)")
    ap.add_argument("--train-period", type=int, default=100, help="mlp-train (C5b): force-matching step every this many MD steps")
    ap.add_argument("--cells", type=int, default=32, help="fcc cells per side (N = 4 cells^3 per rank)")
    ap.add_argument("--lattice", default="fcc", choices=["fcc", "sc"], help="fcc: N = 4 cells^3 (C3, C5); sc: N = cells^3 (C2 = sc 32^3 = 32768)")
    ap.add_argument("--nn", type=int, default=128)
    ap.add_argument("--rcut", type=float, default=3.0)
    ap.add_argument("--rbuff", type=float, default=0.4)
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--check-period", type=int, default=5, help="nlist distance-check period (HOOMD check_period)")
    ap.add_argument("--equil", type=int, default=300, help="untimed relaxation steps (force cap + velocity rescale)")
    ap.add_argument("--settle", type=int, default=100, help="untimed plain NVE steps between the relaxation and the warmup")
    ap.add_argument("--sort", action="store_true", help="enable the stand-in's particle sorter (HOOMD SFCPack analogue; measured: no kernel gain)")
    ap.add_argument("--no-fused", action="store_true", help="skip the extra variants (two-kernel dataflow, tensor-less fused mode)")
    ap.add_argument("--one-kernel", action="store_true",
                    help="eds workload: the whole C4 sweep as one kernel (htf_build_eval_forces2) -- the default since the "
                         "tensor is written with streaming stores; --two-kernel selects build + eval2")
    ap.add_argument("--sync-train", action="store_true", help="mlp-train: run the training step on the MD stream (no overlap)")
    ap.add_argument("--two-kernel", action="store_true",
                    help="headline run with separate build and evaluator kernels (htf_config.fused = 0)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the --cells box itself cut into N slabs (default; the BASELINE metric), "
                         "weak = N such boxes side by side, one per rank (config 5)")
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of --steps steps each; `value` is their median (0 = 5 windows when --steps <= 50, else 1)")
    ap.add_argument("--host-nlist-decision", action="store_true",
                    help="read the neighbor-list distance check back to the host (round-1 behaviour); default: the "
                         "rebuild is gated on the device, the step loop never synchronises")
    ap.add_argument("--no-mlp", action="store_true", help="default N = 1 run: skip the pair-MLP sub-records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--f64", action="store_true",
                    help="HOOMD built in double precision: fp64 positions / velocities / forces on the wire, fp32 pair vectors "
                         "and model arithmetic (the reference casts the fp64 buffer to the model dtype, simmodel.py:226-238)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def count_gpus_sysfs():
    """GPUs of this node from the KFD topology (nodes with SIMDs), without touching the HIP runtime: on ROCm builds without
    amdsmi torch.cuda.device_count() falls through to hipGetDeviceCount, which initialises the GPU in the calling process."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except OSError:
            pass
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    if vis is not None and vis.strip() != "":
        n = min(n, len([x for x in vis.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N rank processes from here -- plain children
    of a parent that has not touched the GPU (never a re-exec) -- and relay rank 0's JSON line."""
    import socket
    import subprocess
    backend = os.environ.get("HTF_BENCH_BACKEND", "nccl")
    ndev = count_gpus_sysfs()  # the launcher parent stays GPU-free for certain: no HIP call, not even a device count
    if backend == "nccl" and ndev < args.gpus:
        print("bench.py --gpus %d: this node shows %d GPU(s).  RCCL wants one device per rank; "
              "HTF_BENCH_BACKEND=gloo rehearses the multi-rank path with the ranks sharing devices."
              % (args.gpus, ndev), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # rank 0's stdout carries the JSON line; the other ranks' stdout (library chatter) goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed_at = None
    while True:
        rcs = [pr.poll() for pr in procs]
        if all(rc is not None for rc in rcs):
            break
        if failed_at is None and any(rc not in (None, 0) for rc in rcs):
            failed_at = time.monotonic()  # a rank died: the others would wait for it in a collective forever
        if failed_at is not None and time.monotonic() - failed_at > 20.0:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()  # our own children, by handle
        time.sleep(0.1)
    rcs = [pr.wait() for pr in procs]
    reader.join(timeout=10)
    out = b"".join(chunks)
    for line in out.decode("utf-8", "replace").splitlines():
        # ONE JSON line on stdout; anything else a library printed there (gloo's connection notes) is passed to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


class Env:
    """What one rank process knows about the job."""


def main():
    args = parse()
    if os.environ.get("HTF_BENCH_WATCHDOG"):
        # a rank that hangs (mismatched collectives ...) dumps every thread's stack and exits instead of holding the box
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["HTF_BENCH_WATCHDOG"]), exit=True)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    E = Env()
    E.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    E.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the evaluator has no CPU path")
    # HTF_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path with several ranks on ONE
    # GPU (RCCL refuses that); messages then bounce through host memory -- not a measurement
    E.backend = backend = os.environ.get("HTF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    E.dev = dev = torch.device("cuda", local_rank)
    E.dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        E.dist = dist

    import hoomd_tf_amd as htf
    from hoomd_tf_amd import standin
    E.htf, E.standin = htf, standin

    if args.workload == "eds":
        if world > 1:
            raise SystemExit("the C4 workload is a single-GPU configuration")
        return run_eds(args, htf, standin, dev)
    if args.workload == "generic-lj":
        run_generic_lj(args, htf, standin, dev)
        return
    if args.workload == "dd-self":
        if world > 1:
            raise SystemExit("dd-self is one rank playing every brick")
        run_dd_self(args, htf, standin, dev)
        return
    if args.workload == "ref-lj256":
        if world > 1:
            raise SystemExit("the reference's own benchmark is a 256-particle, single-device workload")
        return run_ref_lj256(args, htf, standin, dev)
    if args.workload in ("c1", "ex01"):
        if world > 1:
            raise SystemExit("BASELINE configs[0] is a single-device plumbing case")
        return run_small(args, htf, standin, dev)
    headline = args.workload == "lj" and not args.f64 and not args.two_kernel
    E.live = None
    out = run_md(args, E, args.workload, variants=not args.no_fused, cpu=not args.no_cpu_baseline)
    live = E.live          # the LJ run's objects, for the guarded multi-rank section below
    if headline and not args.no_mlp and ((world == 1 and args.cells == 32) or (world > 1 and args.scaling == "strong")):
        # north_star: "LJ AND MLP pair-potential boxes ... at 1/2/4/8": the same C3 system driven by the pair-MLP in its default
        # precision (split16), a bounded number of steps -- on one GPU with the fp32-MFMA and bf16-split evaluators as variants,
        # on N ranks the box cut as the LJ run cuts it (per-rank `roofline`: rank 0's share of the evaluator)
        import copy
        a2 = copy.copy(args)
        # (relaxed as the LJ run relaxes the box -- after 100 steps instead of 300 the liquid holds 3 % more pairs inside the cut-off
        #  and the evaluator has 3 % more tiles to do -- and three windows: ~0.4 s in all at 1 ms per step)
        a2.steps, a2.warmup, a2.equil, a2.windows = min(args.steps, 40), min(args.warmup, 5), args.equil, 3
        sub = run_md(a2, E, "mlp", variants=not args.no_fused and world == 1, cpu=not args.no_cpu_baseline, keep_live=False)
        out["mlp"] = {k: sub[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "kernels", "gpu_state",
                                         "roofline", "fp32_variant", "split_variant", "cpu_baseline", "energy_per_particle", "kT_final") if k in sub}
    if world > 1:
        from benchlib import multirank
        # who ran this: device uuid / PCI bus id per rank and what the process group says about itself
        out["ranks"] = multirank.rank_table(E)
        if live is not None and os.environ.get("HTF_BENCH_GUARDED", "1") != "0":
            def emit(extra):
                print(json.dumps(dict(out, **extra)))
                sys.stdout.flush()
            extra = multirank.guarded_section(E, live, out, emit)
            nat = getattr(live["nl"].domain, "_native", None)
            if nat is not None:
                # what RCCL says about the library's communicator, rank by rank
                mine = multirank.add_native_comm_info({"rank": rank}, nat)
                table = [None] * world
                E.dist.all_gather_object(table, mine)
                for row, t in zip(out["ranks"], table):
                    row["rccl_communicator"] = t.get("rccl_communicator")
            out.update(extra)
            multirank.promote_verified(out)
    if rank == 0:
        print(json.dumps(out))
    if E.dist is not None:
        E.dist.barrier()
        E.dist.destroy_process_group()


if __name__ == "__main__":
    main()
