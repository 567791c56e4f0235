cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_brick.py tests/test_gpu_standin.py -x -q -m gpu > gpurun_out/r5b/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r5b/pytest.log
for g in 8x1x1 4x2x1; do
 for t in local peer native; do
  timeout 300 python bench.py --workload dd-self --grid $g --transport $t --no-cpu-baseline 2>gpurun_out/r5b/dd_${g}_$t.err | grep '^{' | tail -1 > gpurun_out/r5b/dd_${g}_$t.json
  python -c "import json;d=json.load(open('gpurun_out/r5b/dd_${g}_$t.json'));print('$g $t',d['ms_per_step'],d['value'])" || tail -5 gpurun_out/r5b/dd_${g}_$t.err
 done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5b/prof -- python3 bench.py --workload dd-self --grid 8x1x1 --transport local --no-cpu-baseline > gpurun_out/r5b/prof.log 2>&1
find gpurun_out/r5b/prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/r5b/dd_kernel_stats.csv \;
find gpurun_out/r5b/prof -name '*kernel_trace.csv' -exec cp {} gpurun_out/r5b/dd_kernel_trace.csv \;
