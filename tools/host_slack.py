"""How far ahead of the GPU does the host run?  Times the one blocking call of an iteration (gsr_bin_prepare: launches
K3 / the depth sort / K4 and then waits for the pair count) inside bench.py's default run: if the host arrives early it
waits there (slack); if the wait is ~0 the iteration is host-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--repeats", "1", "--render-steps", "0"] + sys.argv[1:]
import bench  # noqa
import diff_gaussian_rasterization as dgr  # noqa

lib = dgr.lib
orig = lib.gsr_bin_prepare
acc = {"n": 0, "t": 0.0, "ts": []}


def timed(*a):
    t0 = time.perf_counter()
    r = orig(*a)
    dt = time.perf_counter() - t0
    acc["n"] += 1
    acc["t"] += dt
    acc["ts"].append(dt)
    return r


lib.gsr_bin_prepare = timed
t0 = time.perf_counter()
bench.main()
ts = sorted(acc["ts"][20:])
print(f"gsr_bin_prepare: {acc['n']} calls, median {ts[len(ts)//2]*1e6:.0f} us, p10 {ts[len(ts)//10]*1e6:.0f} us, p90 {ts[9*len(ts)//10]*1e6:.0f} us "
      f"(its own launches cost ~30 us; the rest is the host waiting for the GPU)", file=sys.stderr)
