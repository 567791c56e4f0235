# Round-6 evidence pass (tools/evidence_pass_r05.sh + what round 6 changed: the fused step A/B, the training sweep's residual windows, the
# multi-rank line's new fields -- rank table, MLP sub-record, guarded native / peer sections -- rehearsed with ranks sharing the GPU): everything DESIGN.md / README.md quote is regenerated here, on one GPU box, into
# gpurun_out/final/; tools/collect_profiles.sh <round> then copies the summaries into profiles/ (tracked).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=gpurun_out/final
mkdir -p $F
timeout 1500 python -m pytest tests -q -m gpu > $F/pytest_gpu.log 2>&1; tail -2 $F/pytest_gpu.log           # default binding: pybind11 (_htf_abi.so)
HTF_BINDING=ctypes timeout 1500 python -m pytest tests -q -m gpu > $F/pytest_gpu_ctypes.log 2>&1; tail -1 $F/pytest_gpu_ctypes.log
python -c "import __graft_entry__ as g; g.smoke()" > $F/smoke.log 2>&1; tail -2 $F/smoke.log
jl() { grep '^{' | tail -1; }
timeout 900 python bench.py --steps 20 --warmup 5 2>$F/bench_lj.err | jl > $F/bench_lj.json          # the driver's own command
timeout 900 python bench.py 2>/dev/null | jl > $F/bench_lj_200.json
timeout 900 python bench.py --f64 --no-mlp 2>/dev/null | jl > $F/bench_lj_f64.json
timeout 900 python bench.py --workload wca --no-mlp 2>/dev/null | jl > $F/bench_wca.json
timeout 900 python bench.py --workload wca --lattice sc --cells 32 2>/dev/null | jl > $F/bench_wca_c2.json
timeout 900 python bench.py --workload mlp --steps 100 --warmup 10 2>/dev/null | jl > $F/bench_mlp.json
timeout 900 python bench.py --workload mlp-fp32 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | jl > $F/bench_mlp_fp32.json
timeout 900 python bench.py --workload mlp-bf16 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | jl > $F/bench_mlp_bf16.json
timeout 900 python bench.py --workload c1 2>/dev/null | jl > $F/bench_c1.json          # BASELINE configs[0], both readings
timeout 900 python bench.py --workload ex01 2>/dev/null | jl > $F/bench_ex01.json
timeout 900 python bench.py --workload mlp-split --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | jl > $F/bench_mlp_split.json
timeout 900 python bench.py --workload mlp-train --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | jl > $F/bench_mlp_train.json
timeout 900 python bench.py --workload eds 2>/dev/null | jl > $F/bench_eds.json
timeout 900 python bench.py --workload eds --f64 --no-cpu-baseline 2>/dev/null | jl > $F/bench_eds_f64.json          # C4 under a HOOMD DOUBLE build
timeout 900 python bench.py --workload generic-lj 2>/dev/null | jl > $F/bench_generic_lj.json          # what leaving the lowered model zoo costs (torch ops + autograd)
HTF_JIT_TAILS=0 timeout 900 python bench.py --workload generic-lj 2>/dev/null | jl > $F/bench_generic_lj_tails0.json          # ... the generated step in its two-row form at every size
HTF_TRAIN_FROM_TENSOR=1 timeout 900 python bench.py --workload generic-lj 2>/dev/null | jl > $F/bench_generic_lj_train_from_tensor.json          # ... the training plan with the tensor written and swept
timeout 120 tools/mlp_mix_probe > $F/mlp_mix_probe.txt 2>&1
# the pair-MLP evaluator's u-form (lever (ii)) against the shipped t-form, same box, twice each (the variant is built by __graft_entry__.build() when it can)
if [ -s build_variants/libhtf_uform.so ]; then for i in 1 2; do for lib in "" build_variants/libhtf_uform.so; do
  HTF_AMD_LIB=$lib HTF_BINDING=ctypes timeout 300 python bench.py --workload mlp --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | jl | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib=%s' % ('$lib' or 'shipped (t-form)'), 'steps/s %.1f' % d['value'], 'evaluator us %.1f' % d['kernels']['eval_forces']['avg_us'])"
done; done > $F/mlp_uform_ab.txt; fi
for c in 16 20 25; do          # the per-rank row counts of the 131k box cut 8 / 4 / 2 ways: inputs of DESIGN 6's predicted scaling table
  timeout 300 python bench.py --cells $c --steps 200 --warmup 20 --no-mlp --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_lj_cells$c.json
done
bash tools/pmc_train.sh > $F/train_sweep_pmc.txt 2>&1          # the training sweep alone (tools/train_probe.py): rocprofv3 duration + SQ counters
timeout 900 python bench.py --workload ref-lj256 2>/dev/null | jl > $F/bench_ref_lj256.json          # the one workload the reference publishes a number for
HTF_BENCH_WATCHDOG=400 HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_rehearsal_2ranks_strong_gloo.json
HTF_BENCH_WATCHDOG=400 HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_rehearsal_8ranks_strong_gloo.json
HTF_BENCH_WATCHDOG=400 HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --scaling weak --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_rehearsal_2ranks_weak_gloo.json
# the driver's own launch form for N > 1 (torch.distributed.run), rehearsed the same way
HTF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_rehearsal_2ranks_strong_torchrun_gloo.json
HTF_BENCH_WATCHDOG=400 HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload mlp --steps 10 --warmup 3 --equil 60 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_rehearsal_2ranks_strong_mlp_gloo.json
# config 5 at its full size (8 x 131072 = 1.05 M particles, force matching on), the 8 ranks sharing this one GPU
HTF_BENCH_WATCHDOG=400 HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --scaling weak --workload mlp-train --train-period 10 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | jl > $F/bench_rehearsal_c5_8ranks_weak_mlptrain_gloo.json
# round 6: the multi-rank line as the driver will run it (ranks sharing this GPU over gloo): rank table, `mlp` sub-record, guarded section
HTF_BENCH_WATCHDOG=500 HTF_BENCH_BACKEND=gloo timeout 700 python bench.py --gpus 8 --grid 4x2x1 --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_rehearsal_8ranks_4x2_strong_gloo.json
HTF_BENCH_WATCHDOG=500 HTF_BENCH_BACKEND=gloo timeout 700 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_rehearsal_4ranks_strong_gloo.json
# round 6: the step as ONE launch (integrator as the force kernel's epilogue) against the separate launches, same box, same command
for v in 0 1; do
  HTF_NO_STEP_EPILOGUE=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-mlp --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_lj_epilogue_off$v.json
  HTF_NO_STEP_EPILOGUE=$v timeout 600 python bench.py --workload wca --lattice sc --cells 32 --steps 100 --windows 5 --no-cpu-baseline --no-fused 2>/dev/null | jl > $F/bench_wca_c2_epilogue_off$v.json
  HTF_NO_STEP_EPILOGUE=$v timeout 300 python bench.py --workload dd-self --grid 8x1x1 --transport local --replan-every 2 --steps 200 2>/dev/null | jl > $F/bench_dd_self_8x1x1_epilogue_off$v.json
  HTF_NO_STEP_EPILOGUE=$v timeout 300 python bench.py --workload dd-self --grid 4x2x1 --transport local --replan-every 2 --steps 200 2>/dev/null | jl > $F/bench_dd_self_4x2x1_epilogue_off$v.json
done
# round 6: the pair-MLP training sweep with one outlier residual among a million rows (residual windows), and its duration
for o in 0 1e5 1e7 1e8; do python tools/train_size_probe.py 64 200 $o 2>&1 | grep -v amdgpu.ids | grep "outlier\|max|g|\|vs fp32"; done > $F/train_size_probe_outlier.txt
python tools/train_probe.py 2>&1 | grep -v amdgpu > $F/train_probe.txt
timeout 300 python tools/soak_nve.py --steps 20000 > $F/soak_nve.json 2>/dev/null          # 20 000 NVE steps at the headline size: energy drift, momentum accounting
timeout 300 python tools/soak_nve.py --steps 20000 --f64 > $F/soak_nve_f64.json 2>/dev/null
# 20 000 steps of the REPLAYED decomposed step as round 6 runs it (integrator + halo pack in the force kernel; `peer`: halo, migration and
# all-reduce without a library): energy, particle count, flags
for t in local peer native; do timeout 300 python tools/soak_brick.py --grid 8x1x1 --transport $t --replan-every 2 2>/dev/null | sed -n '/^{/,/^}/p' > $F/soak_brick_8x1x1_$t.json; done
for t in local peer; do timeout 300 python tools/soak_brick.py --grid 4x2x1 --transport $t --replan-every 2 2>/dev/null | sed -n '/^{/,/^}/p' > $F/soak_brick_4x2x1_$t.json; done
# round 5: one rank's decomposed step at the 8-rank geometries (replica mode), eager and replayed, local delivery and RCCL
timeout 300 python bench.py --workload dd-self --grid 8x1x1 --steps 100 --warmup 20 2>/dev/null | jl > $F/bench_dd_self_8x1x1.json
timeout 300 python bench.py --workload dd-self --grid 4x2x1 --steps 100 --warmup 20 2>/dev/null | jl > $F/bench_dd_self_4x2x1.json
# ... with a re-plan only every second rebuild (BrickDomain(replan_every=2): a ghost layer r_buff thicker, a third graph)
timeout 300 python bench.py --workload dd-self --grid 8x1x1 --replan-every 2 --steps 100 --warmup 20 2>/dev/null | jl > $F/bench_dd_self_8x1x1_replan2.json
timeout 300 python bench.py --workload dd-self --grid 4x2x1 --replan-every 2 --steps 100 --warmup 20 2>/dev/null | jl > $F/bench_dd_self_4x2x1_replan2.json
timeout 200 python tools/rccl_graph_probe.py 2>&1 | cut -c1-190 > $F/rccl_graph_probe.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dd -o dd -- python3 bench.py --workload dd-self --grid 8x1x1 --transport local --steps 200 --warmup 20 --windows 2 > /dev/null 2>&1
find /tmp/p_dd -name "*kernel_stats.csv" -exec cp {} $F/dd_kernel_stats.csv \;
# the traced models' generated kernels by name and duration (htf_jit_rows2_f32_store: the two-row one-kernel step around a generated body)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_gen -o gen -- python3 bench.py --workload generic-lj > /dev/null 2>&1
find /tmp/p_gen -name "*kernel_stats.csv" -exec cp {} $F/generic_lj_kernel_stats.csv \;
HTF_DD_PHASES=1 timeout 300 python bench.py --workload dd-self --grid 8x1x1 --transport local 2>/dev/null | jl > $F/bench_dd_self_phases_8x1x1.json
HTF_DD_PHASES=1 timeout 300 python bench.py --workload dd-self --grid 4x2x1 --transport local 2>/dev/null | jl > $F/bench_dd_self_phases_4x2x1.json
timeout 100 tools/launch_floor_probe > $F/launch_floor_probe.txt 2>&1
# kernel durations: rocprofv3 --kernel-trace --stats of the same commands
Q="--no-cpu-baseline --no-mlp"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_lj -o lj -- python3 bench.py $Q > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mlp -o mlp -- python3 bench.py --workload mlp --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mt -o mt -- python3 bench.py --workload mlp-train --steps 200 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_eds -o eds -- python3 bench.py --workload eds > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -o c2 -- python3 bench.py --workload wca --lattice sc --cells 32 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_f64 -o f64 -- python3 bench.py --f64 $Q > /dev/null 2>&1
for n in lj mlp mt eds c2 f64; do find /tmp/p_$n -name "*kernel_stats.csv" -exec cp {} $F/${n}_kernel_stats.csv \; ; done
# HBM bytes: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one), short runs
S="--no-cpu-baseline --no-mlp --no-fused --steps 20 --warmup 5 --equil 60 --settle 0 --windows 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/c_f -o f -- python3 bench.py $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/c_w -o w -- python3 bench.py $S > /dev/null 2>&1
# where the LJ step's time goes: TA / TCP and SQ counters of its one kernel
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf /tmp/pk_$n
  rocprofv3 --pmc $set --output-format csv -d /tmp/pk_$n -o x -- python3 bench.py $S > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
F = "gpurun_out/final"
def collect(pattern, want=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void htf::") and (want is None or any(w in k for w in want)):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"launches": len(v), "avg": sum(v[len(v) // 2:]) / len(v[len(v) // 2:])} for c, v in d.items()} for k, d in agg.items()}
out = {}
for name, d in (("FETCH_SIZE", "/tmp/c_f"), ("WRITE_SIZE", "/tmp/c_w")):
    c = collect(d + "/**/*counter_collection.csv")
    out[name] = {k: {"launches": v[name]["launches"], "avg_KiB": v[name]["avg"]} for k, v in c.items() if name in v and v[name]["launches"] > 5}
out["_note"] = ("rocprofv3 --pmc (separate passes) of: bench.py --no-cpu-baseline --no-mlp --no-fused --steps 20 --warmup 5 --equil 60 --windows 1; averages over the "
                "second half of each kernel's launches.  gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B/lane stores.")
json.dump(out, open(F + "/pmc_hbm.json", "w"), indent=1)
lj = collect("/tmp/pk_*/**/*counter_collection.csv", want=["fused_forces_tails_kernel<1, true, 4, float", "fused_forces_rows2_kernel<1, true"])
json.dump(lj, open(F + "/pmc_lj_kernel.json", "w"), indent=1)
mlp = collect("/tmp/pm_*/**/*counter_collection.csv", want=["pair_mlp_kernel"])
trn = collect("/tmp/pt_*/**/*counter_collection.csv", want=["mlp_grad", "pair_mlp_kernel"])
json.dump({"evaluator (bench.py --workload mlp, fp32 MFMA + the split variant)": mlp, "training (bench.py --workload mlp-train)": trn,
           "_note": "SQ_INSTS_MFMA: wave-level MFMA instructions; SQ_VALU_MFMA_BUSY_CYCLES: cycles the matrix pipe is busy, summed over SIMDs; "
                    "matrix pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"}, open(F + "/pmc_mlp.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
ls -la $F
