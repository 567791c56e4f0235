# copy the evidence pass's outputs (gpurun_out/final, scratch) into profiles/ (tracked), named per round
R=${1:-r06}
cd "$(dirname "$0")/.."
F=gpurun_out/final
for n in bench_ref_lj256 bench_lj bench_lj_200 bench_lj_f64 bench_wca bench_wca_c2 bench_mlp bench_mlp_fp32 bench_mlp_bf16 bench_mlp_split bench_c1 bench_ex01 bench_mlp_train bench_rehearsal_2ranks_strong_gloo bench_rehearsal_8ranks_strong_gloo bench_rehearsal_2ranks_weak_gloo bench_rehearsal_c5_8ranks_weak_mlptrain_gloo bench_rehearsal_2ranks_strong_torchrun_gloo bench_rehearsal_2ranks_strong_mlp_gloo bench_eds_f64 bench_generic_lj bench_generic_lj_tails0 bench_generic_lj_train_from_tensor bench_lj_cells16 bench_lj_cells20 bench_lj_cells25 bench_dd_self_8x1x1 bench_dd_self_4x2x1 bench_dd_self_8x1x1_replan2 bench_dd_self_4x2x1_replan2 bench_rehearsal_8ranks_4x2_strong_gloo bench_rehearsal_4ranks_strong_gloo bench_lj_epilogue_off0 bench_lj_epilogue_off1 bench_wca_c2_epilogue_off0 bench_wca_c2_epilogue_off1 bench_dd_self_8x1x1_epilogue_off0 bench_dd_self_8x1x1_epilogue_off1 bench_dd_self_4x2x1_epilogue_off0 bench_dd_self_4x2x1_epilogue_off1; do
  [ -s $F/$n.json ] && cp $F/$n.json profiles/${R}_$n.json
done
cp $F/bench_eds.json profiles/${R}_bench_eds_c4.json
cp $F/lj_kernel_stats.csv profiles/${R}_bench_lj_kernel_stats.csv
cp $F/mlp_kernel_stats.csv profiles/${R}_bench_mlp_kernel_stats.csv
cp $F/mt_kernel_stats.csv profiles/${R}_bench_mlp_train_kernel_stats.csv
cp $F/eds_kernel_stats.csv profiles/${R}_bench_eds_c4_kernel_stats.csv
[ -s $F/c2_kernel_stats.csv ] && cp $F/c2_kernel_stats.csv profiles/${R}_bench_wca_c2_kernel_stats.csv
[ -s $F/dd_kernel_stats.csv ] && cp $F/dd_kernel_stats.csv profiles/${R}_bench_dd_self_kernel_stats.csv
[ -s $F/generic_lj_kernel_stats.csv ] && cp $F/generic_lj_kernel_stats.csv profiles/${R}_bench_generic_lj_kernel_stats.csv
for n in bench_dd_self_phases_8x1x1 bench_dd_self_phases_4x2x1; do [ -s $F/$n.json ] && cp $F/$n.json profiles/${R}_$n.json; done
[ -s $F/launch_floor_probe.txt ] && cp $F/launch_floor_probe.txt profiles/${R}_launch_floor_probe.txt
[ -s $F/rccl_graph_probe.txt ] && cp $F/rccl_graph_probe.txt profiles/${R}_rccl_graph_probe.txt
[ -s $F/f64_kernel_stats.csv ] && cp $F/f64_kernel_stats.csv profiles/${R}_bench_lj_f64_kernel_stats.csv
[ -s $F/fetch_calib.json ] && cp $F/fetch_calib.json profiles/${R}_fetch_calib.json
cp $F/pmc_hbm.json profiles/${R}_bench_lj_pmc_hbm.json
cp $F/pmc_lj_kernel.json profiles/${R}_bench_lj_pmc_kernel.json
cp $F/pmc_mlp.json profiles/${R}_bench_mlp_pmc.json
[ -s $F/pmc_c2_c4.json ] && cp $F/pmc_c2_c4.json profiles/${R}_bench_c2_c4_pmc.json
[ -s $F/gather_probe2.txt ] && cp $F/gather_probe2.txt profiles/${R}_gather_probe.txt
[ -s $F/store_probe.txt ] && cp $F/store_probe.txt profiles/${R}_store_probe.txt
for n in valu_cost_probe mlp_mix_probe mlp_uform_ab train_size_probe_outlier train_probe; do [ -s $F/$n.txt ] && cp $F/$n.txt profiles/${R}_$n.txt; done
cp gpurun_out/parity_stats.json profiles/${R}_parity_stats.json
[ -s $F/train_sweep_pmc.txt ] && grep -v "^stats\|amdgpu.ids" $F/train_sweep_pmc.txt > profiles/${R}_train_sweep_pmc.txt
for n in soak_nve soak_nve_f64 soak_brick_8x1x1_local soak_brick_8x1x1_peer soak_brick_8x1x1_native soak_brick_4x2x1_local soak_brick_4x2x1_peer; do [ -s $F/$n.json ] && cp $F/$n.json profiles/${R}_$n.json; done
for n in pytest_gpu pytest_gpu_ctypes smoke; do [ -s $F/$n.log ] && cp $F/$n.log profiles/${R}_$n.log; done
tail -2 $F/pytest_gpu.log; tail -1 $F/smoke.log
