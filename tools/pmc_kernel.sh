# PMC counters for one kernel of one bench command:  bash tools/pmc_kernel.sh <kernel-name-substring> <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=$1; shift
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d /tmp/pk_$n -o x -- python3 bench.py "$@" > /dev/null 2>&1
  python3 - "$n" "$K" <<'PY'
import csv,glob,sys,collections
n,K=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pk_%s/**/*counter_collection.csv"%n, recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
done
