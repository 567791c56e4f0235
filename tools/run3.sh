# round-2 GPU pass 3: full GPU test suite (no -x: collect every strict-bound failure), smoke, default bench,
# 2-rank rehearsals of the strong / weak multi-rank bench on one GPU (gloo), the gather probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r3/pytest_gpu.log 2>&1; tail -3 gpurun_out/r3/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/smoke.log 2>&1; tail -2 gpurun_out/r3/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err; tail -c 600 gpurun_out/r3/bench_default.json; tail -3 gpurun_out/r3/bench_default.err
HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/bench_g2_strong.json 2> gpurun_out/r3/bench_g2_strong.err; echo "rc=$?"; tail -c 400 gpurun_out/r3/bench_g2_strong.json; tail -3 gpurun_out/r3/bench_g2_strong.err
HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --scaling weak --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/bench_g2_weak.json 2> gpurun_out/r3/bench_g2_weak.err; echo "rc=$?"; tail -c 400 gpurun_out/r3/bench_g2_weak.json; tail -3 gpurun_out/r3/bench_g2_weak.err
HTF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-fused > gpurun_out/r3/bench_g8_strong.json 2> gpurun_out/r3/bench_g8_strong.err; echo "rc=$?"; tail -c 400 gpurun_out/r3/bench_g8_strong.json; tail -3 gpurun_out/r3/bench_g8_strong.err
./tools/gather_probe2 > gpurun_out/r3/gather_probe2.txt 2>&1; cat gpurun_out/r3/gather_probe2.txt
