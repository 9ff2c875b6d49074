"""Build libgsraster.so (the C-ABI of include/gsraster.h) for gfx950 with plain hipcc.

No torch headers, no cmake: four translation units compiled in parallel and linked into
grendel-gs_amd/diff_gaussian_rasterization/libgsraster.so (kept in-tree so it travels to the GPU box).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT_DIR = os.path.join(ROOT, "diff_gaussian_rasterization")
SO = os.path.join(OUT_DIR, "libgsraster.so")
UNITS = ["preprocess", "binning", "composite", "loss", "optim", "activations", "knn", "compact", "exchange", "api"]
HEADERS = [os.path.join(HERE, "common.h"), os.path.join(HERE, "radix.h"), os.path.join(HERE, "binning_persist.h"), os.path.join(HERE, "binning_rows.h"), os.path.join(os.path.dirname(ROOT), "include", "gsraster.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    units = [u for u in UNITS if os.path.exists(os.path.join(HERE, u + ".hip"))]
    hipcc = _hipcc()
    extra = os.environ.get("GSR_DEFINES", "").split()  # experiment switches (ablations); empty in production
    if extra:
        force = True

    def compile_one(u):
        src, obj = os.path.join(HERE, u + ".hip"), os.path.join(HERE, u + ".o")
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {u}.hip:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        objs = list(ex.map(compile_one, units))
    if force or _stale(SO, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", SO]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
