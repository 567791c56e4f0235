# Standalone timing + PMC counters of the pair-MLP training sweep:  bash tools/pmc_train.sh [kernel-substring] > gpurun_out/train_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=${1:-mlp_grad_tr16_kernel}
python3 tools/train_probe.py
rm -rf /tmp/pt_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_stats -o t -- python3 tools/train_probe.py > /dev/null 2>&1
find /tmp/pt_stats -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-60,200-400 | sed 's/^/stats: /'
find /tmp/pt_stats -name "*kernel_stats.csv" -exec grep -h "mlp_grad\|pair_mlp_kernel\|resid_max\|reduce_partials" {} \; | awk -F'",' '{n=split($1,a,"("); print a[1], $2, $3, $4}' | cut -c1-200
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pt_$n
  rocprofv3 --pmc $set --output-format csv -d /tmp/pt_$n -o x -- python3 tools/train_probe.py > /dev/null 2>&1
  python3 - "$n" "$K" <<'PY'
import csv,glob,sys,collections
n,K=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pt_%s/**/*counter_collection.csv"%n, recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
done
