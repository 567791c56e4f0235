# round-2 A/B batch 2: counters available + occupancy limit through a dynamic-LDS pad + TA/TCP counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
run() { env "$@" python3 tools/fused_ab.py $ARGS 2>&1 | grep median; }
ARGS="--relax 200"
for pad in 0 20000 26000 32000 40000 53000; do run HTF_FUSED_LDSPAD=$pad; done
ARGS="--relax 200 --fused 1"
for pad in 0 20000 26000 32000 40000 53000; do run HTF_FUSED_LDSPAD=$pad; done
for F in 2 1; do
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pk_$n
  rocprofv3 --pmc $set --output-format csv -d /tmp/pk_$n -o x -- python3 tools/fused_ab.py --relax 200 --fused $F --reps 20 > /dev/null 2>&1
  python3 - "$n" "fused_forces_rows2" "$F" <<'PY'
import csv,glob,sys,collections
n,K,F=sys.argv[1],sys.argv[2],sys.argv[3]
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pk_%s/**/*counter_collection.csv"%n, recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    v=v[len(v)//2:]  # the timed (post-relaxation) launches
    print("fused=%s"%F, k, len(v), sum(v)/len(v))
PY
done
done
