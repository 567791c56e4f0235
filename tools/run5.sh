cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r5/pytest_gpu.log 2>&1; tail -3 gpurun_out/r5/pytest_gpu.log
grep -E "^FAILED|^ERROR" gpurun_out/r5/pytest_gpu.log | head -20
for c in 1 0 1 0; do HTF_FUSED2_COMPACT=$c timeout 300 python bench.py --workload eds > gpurun_out/r5/eds_c$c.json 2>/dev/null; python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r5/eds_c$c.json') if l.startswith('{')][0]); print('eds compact=$c', round(d['value'],1), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()})"; done
for g in 0 8 16 32; do HTF_FUSED_GRID=$g timeout 300 python bench.py --workload wca --lattice sc --cells 32 --no-fused --no-cpu-baseline > gpurun_out/r5/wca_g$g.json 2>/dev/null; python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r5/wca_g$g.json') if l.startswith('{')][0]); print('wca sc32 grid=$g', round(d['value'],1), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3))"; done
