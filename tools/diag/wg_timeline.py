"""Per-workgroup timeline of the composite backward (K10) on ONE rank of a fake W-rank world: when did every (tile,
segment) workgroup start and end (100 MHz wall clock), how many list entries did it walk, on which XCD?  Diagnostic for
the thin-band question of DESIGN.md section 3 ("a band that is one round of workgroups takes 2.5x its share").

Step 1 (build container): python tools/diag/wg_timeline.py build      -> variants/libgsraster_wgtime.so
Step 2 (GPU box):         GSRASTER_LIB=$PWD/variants/libgsraster_wgtime.so python tools/diag/wg_timeline.py run [W]
The instrumented source is generated from csrc/composite.hip by text patches and is not kept in the tree."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build():
    src = open(os.path.join(ROOT, "grendel-gs_amd", "csrc", "composite.hip")).read()
    a = "__device__ unsigned long long g_walked[2];"
    assert a in src
    src = src.replace(a, a + "\n__device__ unsigned long long g_wgt[1 << 16][4];\n__device__ unsigned int g_wgt_n;\n", 1)
    b = "    if (!compute_locally[tile]) return;\n    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n    const int tx = tile % gx, ty = tile / gx;"
    assert src.count(b) == 1
    src = src.replace(b, b + "\n    const unsigned long long t0__ = wall_clock64();", 1)
    c = "        raw = raw_nxt;\n        id_cur = id_nxt;\n        id_nxt = id_nxt2;\n    }\n}\n"
    assert src.count(c) == 1
    rec = ("        raw = raw_nxt;\n        id_cur = id_nxt;\n        id_nxt = id_nxt2;\n    }\n"
           "    if (threadIdx.x == 0) {\n        const unsigned int k__ = atomicAdd(&g_wgt_n, 1u);\n"
           "        if (k__ < (1u << 16)) {\n            g_wgt[k__][0] = t0__;\n            g_wgt[k__][1] = wall_clock64();\n"
           "            g_wgt[k__][2] = ((unsigned long long)tile << 8) | (unsigned long long)sidx;\n"
           "            g_wgt[k__][3] = ((unsigned long long)(bmax - c_lo) << 8) | (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);\n"
           "        }\n    }\n}\n")
    src = src.replace(c, rec, 1)
    d = 'extern "C" int gsr_composite_walked('
    api = ('extern "C" int gsr_debug_wgtime(unsigned long long *out, unsigned int *n, int reset) {\n'
           "    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wgt_n), sizeof(unsigned int)) != hipSuccess) return -1;\n"
           "    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgt), sizeof(unsigned long long) * 4 * (1u << 16)) != hipSuccess) return -1;\n"
           "    if (reset) { unsigned int z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgt_n), &z, sizeof(z)) != hipSuccess) return -1; }\n"
           "    return 0;\n}\n\n")
    src = src.replace(d, api + d, 1)
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    path = os.path.join(ROOT, "variants", "composite_wgtime.hip")
    open(path, "w").write(src)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "wgtime", "--src", f"composite={path}"],
                   check=True)


def run(W):
    for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT, os.path.join(ROOT, "tools")):
        sys.path.insert(0, p)
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    import argparse

    import numpy as np
    import torch

    import bench
    import fake_world_bench as F
    import gaussian_renderer.workload_division as wd
    import utils.general_utils as utils
    import diff_gaussian_rasterization as dgr

    lib = dgr.lib
    fn = lib.gsr_debug_wgtime
    fn.restype = ctypes.c_int
    F.install_fake_collectives()
    wd._gather_times_on_host = lambda mine: [list(mine) for _ in range(utils.DEFAULT_GROUP.size())]
    wd._update_heuristics = lambda *a, **k: None
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rank = W // 2
    a = argparse.Namespace(gaussians=0, width=0, height=0, bsz=0, views=8, opacity_logit_mean=0.0, opacity_logit_std=2.0,
                           device_scene=False, no_priming=False, no_fuse_backward=False, graph="off", balance_every=0)
    os.environ["WORLD_SIZE"] = str(W)
    utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = (rank if W > 1 else 0), 0, W
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = F.FakeGroup(W, rank) if W > 1 else utils.SingleGPUGroup()

    # run the workload for a few steps, then look at the LAST K10 launch: reset the buffer before every backward
    orig = lib.gsr_render_backward_seg
    n = ctypes.c_uint(0)

    def spy(*args):
        torch.cuda.synchronize()
        fn(None, ctypes.byref(n), 1)
        return orig(*args)

    lib.gsr_render_backward_seg = spy
    bench.run_workload(a, "c2", W, rank if W > 1 else 0, dev, 4, 2, 1, 0, single_view=(W == 1), collect_kernels=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (4 * (1 << 16)))()
    fn(buf, ctypes.byref(n), 0)
    k = min(n.value, 1 << 16)
    r = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[:k].astype(np.int64)
    t0, t1 = r[:, 0], r[:, 1]
    base = t0.min()
    start, end = (t0 - base) / 100.0, (t1 - base) / 100.0  # microseconds (100 MHz)
    dur = end - start
    walked, xcd, seg = r[:, 3] >> 8, r[:, 3] & 7, r[:, 2] & 255
    print(f"W={W}: {k} workgroups with work; kernel span {end.max():.1f} us; sum of durations {dur.sum():.0f} us "
          f"(= {dur.sum() / end.max():.0f} workgroups busy on average of 1024 slots)")
    print(f"  duration us: mean {dur.mean():.1f}  p50 {np.percentile(dur, 50):.1f}  p90 {np.percentile(dur, 90):.1f}  "
          f"p99 {np.percentile(dur, 99):.1f}  max {dur.max():.1f}")
    print(f"  start us:    p50 {np.percentile(start, 50):.1f}  p90 {np.percentile(start, 90):.1f}  max {start.max():.1f}")
    print(f"  walked entries: mean {walked.mean():.0f}  p90 {np.percentile(walked, 90):.0f}  max {walked.max()}; segments > 0: {(seg > 0).sum()}")
    print(f"  us per 64 walked entries (p50 / p90): {np.percentile(dur / np.maximum(walked / 64.0, 1), 50):.2f} / "
          f"{np.percentile(dur / np.maximum(walked / 64.0, 1), 90):.2f}")
    for q in (0.25, 0.5, 0.75, 0.9, 1.0):
        tq = end.max() * q
        print(f"  at {tq:6.1f} us: {(start <= tq).sum() - (end <= tq).sum():5d} workgroups running, {(end <= tq).sum():5d} done")
    order = np.argsort(-dur)[:8]
    print("  longest:", [(f"{dur[i]:.0f}us", int(walked[i]), f"start {start[i]:.0f}", f"xcd {xcd[i]}") for i in order])
    print("  per XCD last end:", [f"{end[xcd == x].max():.0f}" for x in range(8) if (xcd == x).any()])


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
