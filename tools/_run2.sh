cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 900 python bench.py --workload generic-lj 2>gpurun_out/r5c/generic.err | grep '^{' | tail -1 > gpurun_out/r5c/bench_generic_lj.json; tail -5 gpurun_out/r5c/generic.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c/bench_generic_lj.json'))
for k,v in d['sizes'].items():
    print(k, {n: round(v[n]['steps_per_s']) for n in v if isinstance(v[n], dict)}, round(v['mixture_over_lowered_lj_time'],3), round(v['torch_over_traced_mixture_time'],1))
PY
