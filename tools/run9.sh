cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r9
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r9/pytest_gpu.log 2>&1; tail -3 gpurun_out/r9/pytest_gpu.log; grep -E "^FAILED" gpurun_out/r9/pytest_gpu.log | head
for T in 4 0 4 0; do HTF_FUSED_TAILS=$T python bench.py --steps 20 --warmup 5 --no-mlp --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r9/b.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r9/b.json').read()); print('lj tails=$T', round(d['value'],1), d['kernels']['build_eval_forces']['avg_us'], 'fused1', round(d['fused_variant']['value'],1), 'tfc', round(d['tfcompute_variant']['value'],1))"; done
for T in 4 0; do HTF_FUSED_TAILS=$T python bench.py --f64 --no-mlp --no-cpu-baseline --no-fused 2>/dev/null | grep '^{' > gpurun_out/r9/b.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r9/b.json').read()); print('f64 tails=$T', round(d['value'],1), d['kernels']['build_eval_forces']['avg_us'])"; done
for T in 4 0; do HTF_FUSED_TAILS=$T python bench.py --workload wca --lattice sc --cells 32 --no-cpu-baseline --no-fused 2>/dev/null | grep '^{' > gpurun_out/r9/b.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r9/b.json').read()); print('c2 tails=$T', round(d['value'],1), d['kernels']['build_eval_forces']['avg_us'])"; done
for T in 4 0; do HTF_FUSED_TAILS=$T python bench.py --workload mlp-train --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r9/b.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r9/b.json').read()); print('mlp-train tails=$T', round(d['value'],1), d['kernels']['build_eval_forces']['avg_us'])"; done
