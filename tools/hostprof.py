"""host-side overhead of one training iteration: cProfile of bench.py's step on a tiny scene (GPU work ~0)"""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--gaussians", "4000", "--width", "128",
            "--height", "128", "--render-steps", "1"]
import bench  # noqa
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
