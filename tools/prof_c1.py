import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--workload", "c1", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
