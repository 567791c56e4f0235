# copy the evidence pass's outputs (gpurun_out/final, scratch) into profiles/ (tracked), named per round
set -e
R=${1:-r01}
cd "$(dirname "$0")/.."
F=gpurun_out/final
cp $F/bench_lj.json profiles/${R}_bench_lj.json
cp $F/bench_lj_f64.json profiles/${R}_bench_lj_f64.json
cp $F/bench_wca.json profiles/${R}_bench_wca.json
cp $F/bench_wca_c2.json profiles/${R}_bench_wca_c2.json
cp $F/bench_mlp.json profiles/${R}_bench_mlp.json
cp $F/bench_mlp_bf16.json profiles/${R}_bench_mlp_bf16.json
cp $F/bench_mlp_split.json profiles/${R}_bench_mlp_split.json
cp $F/bench_mlp_train.json profiles/${R}_bench_mlp_train.json
cp $F/bench_eds.json profiles/${R}_bench_eds_c4.json
cp $F/lj_kernel_stats.csv profiles/${R}_bench_lj_kernel_stats.csv
cp $F/mlp_kernel_stats.csv profiles/${R}_bench_mlp_kernel_stats.csv
cp $F/mt_kernel_stats.csv profiles/${R}_bench_mlp_train_kernel_stats.csv
cp $F/eds_kernel_stats.csv profiles/${R}_bench_eds_c4_kernel_stats.csv
cp $F/pmc_hbm.json profiles/${R}_bench_lj_pmc_hbm.json
cp $F/mfma_valu_probe2.txt profiles/${R}_mfma_valu_probe.txt
cp $F/store_probe.txt profiles/${R}_store_probe.txt
cp gpurun_out/parity_stats.json profiles/${R}_parity_stats.json
tail -2 $F/pytest_gpu.log; cat $F/smoke.log | tail -1
