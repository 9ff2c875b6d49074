"""Per-workgroup phase timeline of the one-sweep radix passes (K4 / K6: csrc/radix.h onesweep_scatter, called by
radix_onesweep_kernel and emit_scatter_kernel) on the bench scene: when does a workgroup of 4 096 pairs start, how long
does it spend loading / decoding its pairs, in the per-wave histogram, in the digit scans, in the look-back, ranking and
writing out?  Diagnostic for "the D-sized passes did not get faster with 19 % fewer pairs" (DESIGN.md section 8).

Step 1 (build container): python tools/diag/radix_timeline.py build   -> variants/libgsraster_rxtime.so
Step 2 (GPU box):         python tools/diag/radix_timeline.py run [n_gaussians]
The instrumented sources are generated from csrc/radix.h and csrc/binning.hip by text patches (variants/, not tracked)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "grendel-gs_amd", "csrc")
NREC, NW = 1 << 15, 10


def _sub(src, old, new, count=1):
    assert src.count(old) == count, (src.count(old), old[:80])
    return src.replace(old, new)


def build(window=None):
    rx = open(os.path.join(CSRC, "radix.h")).read()
    if window:  # (look-back window experiment: python tools/diag/radix_timeline.py build 8)
        rx = _sub(rx, "constexpr int LB_WINDOW = 4;", f"constexpr int LB_WINDOW = {int(window)};")
    rx = _sub(rx, "// LDS of one radix pass workgroup\n",
              f"__device__ unsigned long long g_rxt[{NREC}][{NW}];\n__device__ unsigned int g_rxt_n;\n"
              "// LDS of one radix pass workgroup\n")
    rx = _sub(rx, "uint32_t *__restrict__ vals_out) {\n    constexpr int TILE = ITEMS * THREADS;",
              "uint32_t *__restrict__ vals_out, unsigned long long t_start__ = 0ull) {\n"
              "    const unsigned long long t_in__ = wall_clock64();\n    constexpr int TILE = ITEMS * THREADS;")
    # after the histogram barrier
    rx = _sub(rx, "    __syncthreads();\n    {  // thread d: digit d\n",
              "    __syncthreads();\n    const unsigned long long t_hist__ = wall_clock64();\n"
              "    unsigned long long t_scan__ = 0ull;\n    {  // thread d: digit d\n")
    rx = _sub(rx, "        uint32_t excl = 0;\n        if (live && bid > 0) {",
              "        uint32_t excl = 0;\n        t_scan__ = wall_clock64();\n        if (live && bid > 0) {")
    rx = _sub(rx, "    __syncthreads();\n    const unsigned long long lt = (1ull << lane) - 1ull;",
              "    const unsigned long long t_lb_me__ = wall_clock64();\n    __syncthreads();\n"
              "    const unsigned long long t_lb__ = wall_clock64();\n    const unsigned long long lt = (1ull << lane) - 1ull;")
    rx = _sub(rx, "    __syncthreads();\n    const long long rem = n - bbase;",
              "    const unsigned long long t_rank_me__ = wall_clock64();\n    __syncthreads();\n"
              "    const unsigned long long t_rank__ = wall_clock64();\n    const long long rem = n - bbase;")
    rx = _sub(rx, "            vals_out[dst] = sm.sval[i];\n        }\n    }\n}\n",
              "            vals_out[dst] = sm.sval[i];\n        }\n    }\n"
              "    if (threadIdx.x == 0) {\n        const unsigned int k__ = atomicAdd(&g_rxt_n, 1u);\n"
              f"        if (k__ < {NREC}u) {{\n"
              "            unsigned long long *r__ = g_rxt[k__];\n"
              "            r__[0] = ((unsigned long long)THREADS << 40) | ((unsigned long long)shift << 32) | bid;\n"
              "            r__[1] = t_start__; r__[2] = t_in__; r__[3] = t_hist__; r__[4] = t_scan__; r__[5] = t_lb_me__;\n"
              "            r__[6] = t_lb__; r__[7] = t_rank_me__; r__[8] = t_rank__; r__[9] = wall_clock64();\n"
              "        }\n    }\n}\n")
    rx = _sub(rx, "    const uint32_t bid = onesweep_begin(sm, ticket);\n    if ((long long)bid * (ITEMS * THREADS) >= n) return;"
                  "  // tiles past the end (bounded launches only)",
              "    const unsigned long long t_k0__ = wall_clock64();\n    const uint32_t bid = onesweep_begin(sm, ticket);\n"
              "    if ((long long)bid * (ITEMS * THREADS) >= n) return;")
    rx = _sub(rx, "    onesweep_scatter(sm, key, val, bid, n, shift, nbits, ghist, state, keys_out, vals_out);\n}",
              "    onesweep_scatter(sm, key, val, bid, n, shift, nbits, ghist, state, keys_out, vals_out, t_k0__);\n}")
    bn = open(os.path.join(CSRC, "binning.hip")).read()
    bn = _sub(bn, "    constexpr int WAVES = THREADS / 64;\n    __shared__ OnesweepSmem<ITEMS, THREADS> sm;",
              "    constexpr int WAVES = THREADS / 64;\n    const unsigned long long t_k0__ = wall_clock64();\n"
              "    __shared__ OnesweepSmem<ITEMS, THREADS> sm;")
    bn = _sub(bn, "    onesweep_scatter(sm, key, val, bid, D, 0, xbits, ghist, state, keys_out, vals_out);",
              "    onesweep_scatter(sm, key, val, bid, D, 0, xbits, ghist, state, keys_out, vals_out, t_k0__ | (1ull << 63));")
    bn = _sub(bn, 'extern "C" int gsr_set_depth_tie_order(int mode) {',
              'extern "C" int gsr_debug_rxtime(unsigned long long *out, unsigned int *n, int reset) {\n'
              "    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_rxt_n), sizeof(unsigned int)) != hipSuccess) return -1;\n"
              f"    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rxt), sizeof(unsigned long long) * {NW} * {NREC}) != hipSuccess) return -1;\n"
              "    if (reset) { unsigned int z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rxt_n), &z, sizeof(z)) != hipSuccess) return -1; }\n"
              "    return 0;\n}\n\n"
              'extern "C" int gsr_set_depth_tie_order(int mode) {')
    vd = os.path.join(ROOT, "variants")
    os.makedirs(vd, exist_ok=True)
    open(os.path.join(vd, "radix.h"), "w").write(rx)
    path = os.path.join(vd, "binning_rxtime.hip")
    open(path, "w").write(bn)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "rxtime", "--src", f"binning={path}"],
                   check=True)
    os.remove(os.path.join(vd, "radix.h"))  # (other variant builds must see the production header)


def run(n):
    for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT, os.path.join(ROOT, "tools")):
        sys.path.insert(0, p)
    import math

    import numpy as np
    import torch

    import diff_gaussian_rasterization as dgr
    import synthetic_scene as S
    from diff_gaussian_rasterization import _lib

    lib = ctypes.CDLL(os.path.join(ROOT, "variants", "libgsraster_rxtime.so"))
    for name in ("gsr_bin_prepare_bytes", "gsr_bin_prepare", "gsr_bin_sort_bytes", "gsr_bin_sort"):
        res, args = _lib.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    dev = torch.device("cuda:0")
    W, H = 1920, 1080
    g = S.make_gaussians(n, W, H, seed=0, device=dev)
    cam = S.orbit_cameras(8, W, H, device=dev)[0]
    rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                           torch.zeros(3, device=dev), 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, 3, cam.camera_center, False, False)
    with torch.no_grad():
        m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(
            g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
    P = m2.shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    mask = torch.ones(gy, gx, dtype=torch.bool, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ranges = torch.empty((gx * gy + 1, 2), dtype=torch.int32, device=dev)
    nb = lib.gsr_bin_prepare_bytes(P, W, H)
    prep = torch.empty(nb, dtype=torch.uint8, device=dev)
    D = ctypes.c_int64(0)
    cnt = ctypes.c_uint(0)
    out = np.zeros((NREC, NW), dtype=np.uint64)
    for it in range(4):
        if it == 3:
            torch.cuda.synchronize()
            assert lib.gsr_debug_rxtime(None, ctypes.byref(cnt), 1) == 0
        assert lib.gsr_bin_prepare(P, W, H, ptr(m2), ptr(depths), ptr(radii), ptr(co), ptr(mask), ptr(prep), nb,
                                   ctypes.byref(D), stream) == 0
        sb = lib.gsr_bin_sort_bytes(P, D.value, W, H)
        if it == 0:
            scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
            plist = torch.empty(D.value, dtype=torch.int32, device=dev)
        assert lib.gsr_bin_sort(P, W, H, ptr(mask), ptr(prep), D.value, ptr(scratch), sb, ptr(plist), ptr(ranges),
                                stream) == 0
        torch.cuda.synchronize()
    assert lib.gsr_debug_rxtime(out.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), ctypes.byref(cnt), 0) == 0
    k = min(cnt.value, NREC)
    r = out[:k].astype(np.int64)
    print(f"{n} Gaussians, D = {D.value} pairs, {cnt.value} workgroup records (100 MHz clock: 10 ns ticks)")
    tag = r[:, 0] >> 32
    emit = (out[:k, 1] >> np.uint64(63)).astype(bool)
    r[:, 1] &= (1 << 62) - 1
    names = ["begin -> pairs in registers (load / decode)", "histogram + barrier", "scans + histogram loads",
             "look-back (thread 0)", "wait for the other digits' look-back", "ranking (thread 0)", "wait for the other waves",
             "write-out"]
    for t in sorted(set(zip(tag.tolist(), emit.tolist()))):
        sel = (tag == t[0]) & (emit == t[1])
        x = r[sel]
        order = np.argsort(x[:, 1])
        x = x[order]
        t0 = x[:, 1].min()
        span = (x[:, 9].max() - t0) / 100.0
        kind = "emit_scatter" if t[1] else "radix_onesweep"
        print(f"\n{kind}<{t[0] >> 8} threads> shift {t[0] & 255}: {sel.sum()} workgroups, first start to last end {span:.1f} us")
        dur = np.diff(x[:, 1:], axis=1) / 100.0
        tot = (x[:, 9] - x[:, 1]) / 100.0
        print(f"    workgroup lifetime: mean {tot.mean():.2f} us, p10 {np.percentile(tot, 10):.2f}, p50 {np.percentile(tot, 50):.2f}, "
              f"p90 {np.percentile(tot, 90):.2f}, max {tot.max():.2f}")
        for i, nm in enumerate(names):
            print(f"    {nm:48s} mean {dur[:, i].mean():6.2f} us   p50 {np.percentile(dur[:, i], 50):6.2f}   p90 {np.percentile(dur[:, i], 90):6.2f}")
        st = (x[:, 1] - t0) / 100.0
        qs = [0, 10, 25, 50, 75, 90, 100]
        print("    start times (us after the first): " + "  ".join(f"p{q} {np.percentile(st, q):.1f}" for q in qs))
        if "--rows" in sys.argv and (t[0] & 255) in (0, 7) and (t[0] >> 8) == 512:
            bid = x[:, 0] & 0xFFFFFFFF
            byb = x[np.argsort(bid)]
            base = byb[2000, 1]
            print("    bid: start | aggregate published ~ | look-back start | look-back end (us, relative to bid 2000's start)")
            for row in byb[2000:2048]:
                print(f"    {int(row[0] & 0xFFFFFFFF):5d}: {(row[1] - base) / 100.0:7.2f} | {(row[3] - base) / 100.0:7.2f} | "
                      f"{(row[4] - base) / 100.0:7.2f} | {(row[5] - base) / 100.0:7.2f}   LB {(row[5] - row[4]) / 100.0:5.2f}")
        # workgroups in flight over time
        ev = np.concatenate([np.stack([x[:, 1], np.ones(len(x))], 1), np.stack([x[:, 9], -np.ones(len(x))], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        infl = np.cumsum(ev[:, 1])
        print(f"    workgroups in flight: max {int(infl.max())}, time-weighted mean "
              f"{float((infl[:-1] * np.diff(ev[:, 0])).sum() / max(ev[-1, 0] - ev[0, 0], 1)):.0f}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1_000_000)
