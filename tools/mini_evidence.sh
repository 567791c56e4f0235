cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=gpurun_out/final; mkdir -p $F
timeout 900 python -m pytest tests -q -m gpu > $F/pytest_gpu.log 2>&1; tail -1 $F/pytest_gpu.log
HTF_BINDING=ctypes timeout 900 python -m pytest tests -q -m gpu > $F/pytest_gpu_ctypes.log 2>&1; tail -1 $F/pytest_gpu_ctypes.log
python -c "import __graft_entry__ as g; g.smoke()" > $F/smoke.log 2>&1; tail -1 $F/smoke.log
timeout 900 python bench.py --workload mlp-train --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $F/bench_mlp_train.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mt -o mt -- python3 bench.py --workload mlp-train --steps 200 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
find /tmp/p_mt -name "*kernel_stats.csv" -exec cp {} $F/mt_kernel_stats.csv \;
HTF_BENCH_WATCHDOG=400 HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --scaling weak --workload mlp-train --train-period 10 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $F/bench_rehearsal_c5_8ranks_weak_mlptrain_gloo.json
python -c "
import json; d=json.load(open('$F/bench_mlp_train.json')); print(d['value'], d['kernels']['train_step']['avg_ms'])"
