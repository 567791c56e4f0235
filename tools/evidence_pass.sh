set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.log 2>&1; tail -2 gpurun_out/final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
python bench.py > gpurun_out/final/bench_lj.json 2>gpurun_out/final/bench_lj.err
python bench.py --f64 > gpurun_out/final/bench_lj_f64.json 2>/dev/null
python bench.py --workload wca > gpurun_out/final/bench_wca.json 2>/dev/null
python bench.py --workload wca --cells 20 > gpurun_out/final/bench_wca_c2.json 2>/dev/null
python bench.py --workload mlp --steps 100 --warmup 10 > gpurun_out/final/bench_mlp.json 2>/dev/null
python bench.py --workload mlp-bf16 --steps 100 --warmup 10 > gpurun_out/final/bench_mlp_bf16.json 2>/dev/null
python bench.py --workload mlp-split --steps 100 --warmup 10 > gpurun_out/final/bench_mlp_split.json 2>/dev/null
./tools/mfma_valu_probe2 > gpurun_out/final/mfma_valu_probe2.txt 2>&1
./tools/store_probe > gpurun_out/final/store_probe.txt 2>&1
python bench.py --workload mlp-train --steps 400 --warmup 20 > gpurun_out/final/bench_mlp_train.json 2>/dev/null
python bench.py --workload eds > gpurun_out/final/bench_eds.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_lj -o lj -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mlp -o mlp -- python3 bench.py --workload mlp --steps 50 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mt -o mt -- python3 bench.py --workload mlp-train --steps 200 --warmup 10 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_eds -o eds -- python3 bench.py --workload eds > /dev/null 2>&1
for n in lj mlp mt eds; do find /tmp/p_$n -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/${n}_kernel_stats.csv \; ; done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/c_f -o f -- python3 bench.py --no-cpu-baseline --no-fused --steps 50 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/c_w -o w -- python3 bench.py --no-cpu-baseline --no-fused --steps 50 --warmup 5 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,json,collections
out={}
for name,d in (("FETCH_SIZE","/tmp/c_f"),("WRITE_SIZE","/tmp/c_w")):
    fs=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    agg=collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name")==name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out[name]={k.split("(")[0]:{"launches":len(v),"avg_KiB":sum(v)/len(v)} for k,v in agg.items() if k.startswith("void htf::") and len(v)>5}
out["_note"]="rocprofv3 --pmc (separate passes), bench.py --steps 50 --warmup 5 (tools/evidence_pass.sh). gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM); WRITE_SIZE exact for 16-B/lane stores."
json.dump(out,open("gpurun_out/final/pmc_hbm.json","w"),indent=1)
print(json.dumps(out)[:1500])
PY
ls -la gpurun_out/final
