cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d /tmp/pm_$n -o x -- python3 tools/mlp_ab.py split --reps 3 > /dev/null 2>&1
  python3 - "$n" <<'PY'
import csv,glob,sys,collections
n=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pm_%s/**/*counter_collection.csv"%n, recursive=True):
    for r in csv.DictReader(open(f)):
        if "pair_mlp_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
done
