cd $GRAFT_REPO_ROOT
O=gpurun_out/c4_order.txt; : > $O
for o in lattice morton cells random; do
  ORDER=$o timeout 120 python tools/fused2_ab.py 2>&1 | grep "tensor=" | tr '\n' ' ' >> $O; echo >> $O
done
ORDER=morton CW=3.4 timeout 120 python tools/fused2_ab.py 2>&1 | grep "tensor=" | tr '\n' ' ' >> $O; echo >> $O
for o in lattice sorted shuffled; do timeout 200 python tools/fused_ab.py --order $o 2>&1 | tail -1 | cut -c60- >> $O; done
cat $O
