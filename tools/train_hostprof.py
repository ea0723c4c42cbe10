import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["train_bench.py", "--rays", "1024", "--steps", "20"]
import tools.train_bench as tb
pr = cProfile.Profile()
pr.enable()
tb.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
