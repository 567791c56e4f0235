# time the stand-in's neighbor search kernel inside a bench run (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nl$i -o x -- python3 bench.py --no-cpu-baseline --no-fused --steps 100 > /tmp/b_nl.json 2>/dev/null
  echo "$(tail -1 /tmp/b_nl.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["energy_per_particle"])') nlist_us $(find /tmp/p_nl$i -name '*kernel_stats.csv' -exec grep build_nlist {} \; | awk -F, '{print $(NF-4)}')"
  rm -rf /tmp/p_nl$i
done
