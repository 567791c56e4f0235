# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/fetch_calib.hip), plus the same raw counters on the
# one-kernel LJ step of bench.py.  Writes gpurun_out/fetch_calib.json.  One --pmc set per pass, no trace domains.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/fetch_calib 6 > gpurun_out/fetch_calib_known.json || exit 1
S="--no-cpu-baseline --no-mlp --no-fused --steps 20 --warmup 5 --equil 60 --settle 0 --windows 1"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_MISS_sum TCC_READ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/fc_$i /tmp/fb_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/fc_$i -o x -- ./tools/fetch_calib 6 > /dev/null 2>&1
  rocprofv3 --pmc $set --output-format csv -d /tmp/fb_$i -o x -- python3 bench.py $S > /dev/null 2>&1
done
python3 tools/fetch_calib_report.py > gpurun_out/fetch_calib.json
cat gpurun_out/fetch_calib.json | head -c 6000
