cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_standin.py tests/test_gpu_tensorflow.py -q -m gpu > gpurun_out/r6/pytest.log 2>&1; tail -3 gpurun_out/r6/pytest.log
python bench.py --steps 20 --warmup 5 --no-mlp --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r6/bench.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r6/bench.json').read()); print('lj', d['value'], d['windows_ms_per_step'], d['tfcompute_variant']['value'])"
python bench.py --workload eds 2>/dev/null | grep '^{' > gpurun_out/r6/eds.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r6/eds.json').read()); print('eds', d['value'])"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_lj -o lj -- python3 bench.py --no-cpu-baseline --no-mlp --no-fused > /dev/null 2>&1
find /tmp/p_lj -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6/lj_kernel_stats.csv \;
grep -E "cell_scan|build_nlist|fused_forces_rows2" gpurun_out/r6/lj_kernel_stats.csv | cut -d, -f1-7 | cut -c1-60,200-
