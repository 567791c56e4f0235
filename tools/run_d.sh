cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04_pytest_d.log 2>&1; tail -6 gpurun_out/r04_pytest_d.log
jl() { grep '^{' | tail -1; }
echo "== C4 A/B (old single table vs 8 lane-striped replicas), same box"
for rep in 1 2; do for v in c4old c4rep; do
  HTF_AMD_LIB=build_variants/libhtf_$v.so timeout 300 python bench.py --workload eds --no-cpu-baseline 2>/dev/null | jl | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['value'],1), {n:round(v['avg_us'],2) for n,v in k.items() if 'avg_us' in v})"
done; done
echo "== per-rank sizes of the 131k box cut 2 / 4 / 8 ways (rows per rank), one GPU each"
for c in 16 20 25 32; do
  timeout 300 python bench.py --cells $c --steps 200 --warmup 20 --no-mlp --no-cpu-baseline --no-fused 2>/dev/null | jl > gpurun_out/r04_bench_lj_cells$c.json
  python -c "import json; d=json.load(open('gpurun_out/r04_bench_lj_cells$c.json')); print($c, d['config']['particles_rank0'], round(d['value'],1), round(d['ms_per_step']*1e3,2), {n:round(v['avg_us'],2) for n,v in d['kernels'].items() if 'avg_us' in v}, d['config']['nlist_rebuilds_per_window'])"
done
