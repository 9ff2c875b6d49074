"""-m gpu parity tests: HIP path (through the C-ABI, via the operator mirror) vs the CPU oracles.

Tolerances (north_star: 1e-4 relative fp32):
  * integer / index outputs (radii, point_list, ranges, n_contrib): exact, except that a handful of
    pixels may flip at the hard thresholds (alpha < 1/255, T < 1e-4) because exp() differs in the
    last ulp between CPU libm and v_exp_f32 -- bounded as a FRACTION of pixels.
  * floats: norm-wise relative error <= 1e-4.
"""
import math

import pytest
import torch

import diff_gaussian_rasterization as dgr
import synthetic_scene as S
from helpers import KEYS, cam_kwargs, elem_excess, frac_bad, oracle_c_chain, rel_err, settings_from

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _gpu(g, dev):
    return {k: v.to(dev) for k, v in g.items()}


def _full_mask(cam, dev="cpu"):
    gx, gy = (cam.image_width + 15) // 16, (cam.image_height + 15) // 16
    return torch.ones(gy, gx, dtype=torch.bool, device=dev)


SCENES = [
    # (N, W, H, scale_coef, seed, camera index of 4 orbit views)
    (2000, 200, 120, 0.01, 3, 1),
    (10000, 256, 256, 0.004, 0, 0),      # BASELINE config[0] shape
    (5000, 333, 211, 0.02, 7, 2),        # non-multiple-of-16 image, big splats, long lists
    (20000, 979, 546, 0.006, 5, 3),      # BASELINE config[3] image shape (62 x 35 tiles, ragged right / bottom edge)
    (4000, 80, 4128, 0.01, 9, 0),        # 258 tile rows: beyond the (row, column)-key path -> generic tile-id sort
    (3000, 4096, 64, 0.02, 4, 0),        # exactly 256 tile columns: the widest frame of the (row, column) / row-major paths
]


@pytest.mark.parametrize("N,W,H,sc,seed,ci", SCENES)
@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3])  # training walks 0 -> 1 -> 2 -> 3 (scene/gaussian_model.py:136)
def test_preprocess_forward_matches_c_oracle(device, N, W, H, sc, seed, ci, sh_degree):
    from diff_gaussian_rasterization import GaussianRasterizer
    from oracle import cref as C

    g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
    cam = S.orbit_cameras(4, W, H)[ci]
    ref = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam, sh_degree))
    rast = GaussianRasterizer(settings_from(cam, torch.zeros(3), sh_degree))
    gg = _gpu(g, device)
    m2, rgb, co, radii, depths = rast.preprocess_gaussians(gg["means3D"], gg["scales"], gg["rotations"], gg["shs"],
                                                           gg["opacities"], {})
    rm2, rrgb, rco, rradii, rdepths = ref[:5]
    assert (radii.cpu() != rradii).sum().item() <= max(1, N // 100000), "radii must match (FP contraction off)"
    same = (radii.cpu() == rradii)
    assert rel_err(m2.cpu()[same], rm2[same]) < 1e-6
    assert rel_err(depths.cpu()[same], rdepths[same]) < 1e-6
    assert rel_err(co.cpu()[same], rco[same]) < 1e-5
    assert rel_err(rgb.cpu()[same], rrgb[same]) < 1e-5
    # culled Gaussians are all-zero
    culled = radii == 0
    assert m2[culled].abs().sum().item() == 0 and co[culled].abs().sum().item() == 0


@pytest.mark.parametrize("cull,rows", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("N,W,H,sc,seed,ci", SCENES)
def test_binning_is_ordered_subsequence_of_reference_lists(device, monkeypatch, N, W, H, sc, seed, ci, cull, rows):
    """per tile: the HIP list is a SUBSEQUENCE of the reference-order list (depth, then index), and
    every Gaussian dropped from a tile has alpha < 1/255 on all of the tile's pixels (float64 check),
    i.e. the reference algorithm would have skipped it on every pixel -- with the bounding rect of the
    alpha >= 1/255 ellipse, with the exact per-row tile spans (gsr_set_tile_cull), and (round 6) through the row-major
    pipeline (csrc/binning_rows.h: sorted row segments, one pass over the pairs), which production takes from 5 M pairs
    on and GSR_BIN_ROWS_MIN=1 forces on these small scenes"""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import bin_gaussians
    from oracle import cref as C

    monkeypatch.setenv("GSR_BIN_ROWS_MIN", "1" if rows else "1000000000000")
    dgr.set_tile_cull(cull)
    try:
        _binning_subsequence_case(device, N, W, H, sc, seed, ci, cull, bin_gaussians, C)
    finally:
        dgr.set_tile_cull("env")


def _binning_subsequence_case(device, N, W, H, sc, seed, ci, cull, bin_gaussians, C):

    g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
    cam = S.orbit_cameras(4, W, H)[ci]
    m2, rgb, co, radii, depths, _, _ = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam))
    mask = _full_mask(cam)
    mask[1:3, :] = False  # a non-local band
    mask[-1, ::2] = False  # and a ragged pattern (general masks are allowed by the API)
    pl_ref, ranges_ref, _ = C.bin_and_sort(m2, radii, depths, mask, W, H)
    pl, ranges, D = bin_gaussians(m2.to(device), depths.to(device), radii.to(device), co.to(device),
                                  mask.view(-1).to(torch.uint8).to(device), W, H)
    pl, ranges = pl.cpu().long(), ranges.cpu()
    gx = (W + 15) // 16
    dropped_total, kept_total = 0, 0
    m2d, cod = m2.double(), co.double()
    for t in range(ranges.shape[0]):
        rs, re = int(ranges_ref[t, 0]), int(ranges_ref[t, 1])
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        if not bool(mask.view(-1)[t]):
            assert e - s == 0 and re - rs == 0
            continue
        ref_list, mine = pl_ref[rs:re].long(), pl[s:e]
        keep = torch.isin(ref_list, mine)
        assert torch.equal(ref_list[keep], mine), f"tile {t}: not an order-preserving subsequence"
        drop = ref_list[~keep]
        dropped_total += drop.numel()
        kept_total += mine.numel()
        if drop.numel():
            ty, tx = divmod(t, gx)
            py, px = torch.meshgrid(torch.arange(ty * 16, min(ty * 16 + 16, H)),
                                    torch.arange(tx * 16, min(tx * 16 + 16, W)), indexing="ij")
            dx = m2d[drop, 0:1] - px.reshape(1, -1).double()
            dy = m2d[drop, 1:2] - py.reshape(1, -1).double()
            power = -0.5 * (cod[drop, 0:1] * dx * dx + cod[drop, 2:3] * dy * dy) - cod[drop, 1:2] * dx * dy
            alpha = cod[drop, 3:4] * torch.exp(power)
            alpha = torch.where(power > 0, torch.zeros_like(alpha), alpha)
            assert alpha.max().item() < 1.0 / 255.0, f"tile {t}: a dropped Gaussian would have contributed"
    assert kept_total > 0
    print(f"cull={cull}: pairs kept {kept_total}, dropped {dropped_total} of {pl_ref.numel()}")


def test_speculative_sort_equals_the_sized_sort(device):
    """gsr_bin_sort_bounded (sort launched for the scratch's capacity, pair count read on the device) gives bit-identical
    lists and ranges to gsr_bin_sort with the count read back first -- when the count fits the capacity, when it is far
    below it, and (through the fall-back) when it outgrows it; also with an empty view"""
    import diff_gaussian_rasterization as dgr
    from oracle import cref as C

    W, H = 333, 211
    cam = S.orbit_cameras(4, W, H)[1]

    def inputs(N, sc, seed):
        g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
        m2, rgb, co, radii, depths, _, _ = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam))
        mask = _full_mask(cam)
        mask[2, :] = False
        return [t.to(device) for t in (m2, depths, radii, co, mask.view(-1).to(torch.uint8))]

    small, large, larger = inputs(3000, 0.02, 5), inputs(9000, 0.03, 6), inputs(9000, 0.06, 7)
    empty = [t.clone() for t in small]
    empty[2].zero_()  # no Gaussian is visible: D = 0
    ref = {}
    dgr.release_workspaces()
    dgr.set_speculative_sort(False)
    try:
        for name, x in (("small", small), ("large", large), ("larger", larger), ("empty", empty)):
            pl, rg, D = dgr.bin_gaussians(*x, W, H)
            ref[name] = (pl.clone(), rg.clone(), D)
        assert ref["small"][2] < ref["large"][2] < ref["larger"][2] and ref["empty"][2] == 0
        dgr.release_workspaces()
        dgr.set_speculative_sort(True)
        # first call sizes the scratch (no capacity yet); then: fits / far below / outgrows (falls back, grows) / empty / fits
        for name in ("large", "large", "small", "larger", "empty", "large", "larger"):
            x = {"small": small, "large": large, "larger": larger, "empty": empty}[name]
            pl, rg, D = dgr.bin_gaussians(*x, W, H)
            assert D == ref[name][2], name
            assert torch.equal(rg, ref[name][1]), name
            if D:
                assert pl.numel() == D and torch.equal(pl, ref[name][0]), name
    finally:
        dgr.set_speculative_sort(True)
        dgr.release_workspaces()


@pytest.mark.parametrize("grid_p,grid_s,owners,max_tpw,cull", [
    (None, None, None, None, False), (1, 3, None, 8, False), (2, 5, 1, 8, True), (3, 64, 2, 8, False),
    (2, None, None, None, False), (None, None, None, None, True)])
def test_persistent_binning_equals_the_lookback_pipeline(device, monkeypatch, grid_p, grid_s, owners, max_tpw, cull):
    """K3-K7 as two persistent launches with grid barriers (csrc/binning_persist.h) against the nine launches of the
    look-back pipeline: bit-identical lists and ranges -- with one tile per workgroup (4096 or 8192 elements) and (grids
    capped through the test hooks) many tiles per workgroup, with the chunk-owner table and with the per-chunk search, through the sized
    and the speculative (bounded) sort, with an empty view; each half of the pipeline also combined with the other
    pipeline's half"""
    import diff_gaussian_rasterization as dgr
    from oracle import cref as C

    # (grid_p = 2 without max_tpw: 9000 Gaussians take the 8192-element tiles, 20000 fall back to the look-back pipeline;
    # GSR_BIN_PERSIST_MAXD: the sort kernel whatever the pair count)
    monkeypatch.setenv("GSR_BIN_PERSIST_MAXD", "1000000000")
    monkeypatch.setenv("GSR_BIN_ROWS_MIN", "1")  # (the row-major pipeline whatever the pair count)
    for name, v in (("GSR_BIN_GRID_P", grid_p), ("GSR_BIN_GRID_S", grid_s), ("GSR_BIN_OWNERS", owners),
                    ("GSR_BIN_MAX_TPW", max_tpw)):
        if v is not None:
            monkeypatch.setenv(name, str(v))
    W, H = 333, 211
    cam = S.orbit_cameras(4, W, H)[1]

    def inputs(N, sc, seed, band=None):
        g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
        m2, rgb, co, radii, depths, _, _ = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam))
        mask = _full_mask(cam)
        if band is None:
            mask[2, :] = False
            mask[-1, ::2] = False
        else:
            mask[:band[0], :] = False
            mask[band[1]:, :] = False
        return [t.to(device) for t in (m2, depths, radii, co, mask.view(-1).to(torch.uint8))]

    views = {"small": inputs(3000, 0.02, 5), "large": inputs(9000, 0.03, 6), "larger": inputs(20000, 0.06, 7),
             "band": inputs(9000, 0.05, 8, band=(4, 9))}
    views["empty"] = [t.clone() for t in views["small"]]
    views["empty"][2].zero_()
    order = ("large", "large", "small", "larger", "empty", "band", "large", "larger")
    got = {}
    dgr.set_tile_cull(cull)  # (exact tile spans: both pipelines' K3 and emissions, the sort kernel's rect-derived counts)
    try:
        # (round 6) ... and the row-major pipeline (csrc/binning_rows.h: segments sorted by row, ONE pass over the pairs)
        # behind either prepare step -- with exact tile spans on it hands over to the two-pass pipelines by itself
        for rows in (False, True):
            dgr.set_bin_rowmajor(rows)
            for mode in ("off", "both", "prepare", "sort"):
                if rows and mode in ("both", "sort"):
                    continue  # (the persistent SORT kernel is not reached while the row-major pipeline applies)
                dgr.set_bin_persistent(mode)
                for spec in (False, True):
                    dgr.release_workspaces()
                    dgr.set_speculative_sort(spec)
                    for i, name in enumerate(order):
                        pl, rg, D = dgr.bin_gaussians(*views[name], W, H)
                        got[(rows, mode, spec, i)] = (pl.clone(), rg.clone(), D)
    finally:
        dgr.set_bin_persistent("env")
        dgr.set_bin_rowmajor("env")
        dgr.set_tile_cull("env")
        dgr.set_speculative_sort(True)
        dgr.release_workspaces()
    assert got[(False, "off", False, 3)][2] > got[(False, "off", False, 0)][2] > got[(False, "off", False, 2)][2] > 0
    for (rows, mode, spec, i), (pl, rg, D) in got.items():
        rpl, rrg, rD = got[(False, "off", False, i)]
        assert D == rD, (rows, mode, spec, order[i])
        assert torch.equal(rg, rrg), (rows, mode, spec, order[i])
        if D:
            assert pl.numel() == D and torch.equal(pl, rpl), (rows, mode, spec, order[i])


def test_late_pair_count_equals_the_polled_path_and_survives_an_overflow(device):
    """Round 6 (verdict r05 item 1): the render op launches K3-K7 AND K8 against the capacity of the kept sort scratch and
    looks at the pair count afterwards -- at once, or when the caller settles the counts it collected through
    cuda_args["_gsr_pending"] (gaussian_renderer.render_final does, after its last camera).  Image and n_contrib must be
    BITWISE those of the reference's order (read the count, size the buffers, sort, composite), the gradients equal up to
    the order of K10's atomic adds --
    when the count fits the capacity, and when a view outgrows it: K8 then drew the background from empty ranges, and
    settling repeats the tile sort and K8 in place, after a later view has already been enqueued on the stream."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer

    W, H = 333, 211
    cam = S.orbit_cameras(4, W, H)[1]
    bg = torch.tensor([0.3, 0.2, 0.1])
    mask = _full_mask(cam).to(device)
    wgt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(device)
    scenes = {"small": S.make_gaussians(3000, W, H, seed=5, scale_coef=0.02),
              "large": S.make_gaussians(9000, W, H, seed=6, scale_coef=0.05)}

    def run(name, collector):
        g = scenes[name]
        rast = GaussianRasterizer(settings_from(cam, bg))
        gg = {k: v.to(device).requires_grad_(True) for k, v in g.items()}
        a, b, c, d, e = rast.preprocess_gaussians(*[gg[k] for k in KEYS], {})
        ca = {"stats_collector": {}}
        if collector is not None:
            ca["_gsr_pending"] = collector
        img, n_render, _, nc = rast.render_gaussians(a, c, b, e, d, mask, None, ca)
        return gg, img, nc, n_render

    def finish(gg, img, nc):
        (img * wgt).sum().backward()
        return img.detach().clone(), nc.clone(), {k: gg[k].grad.detach().clone() for k in KEYS}

    try:
        dgr.release_workspaces()
        dgr.set_speculative_sort(False)
        ref, pairs, noise = {}, {}, {}
        for name in scenes:
            gg, img, nc, n_render = run(name, None)
            ref[name], pairs[name] = finish(gg, img, nc), n_render
            # K10 adds a Gaussian's tiles with float atomics in whatever order they finish: the SAME path run twice
            # gives the yardstick for "equal gradients"
            noise[name] = {k: 0.0 for k in KEYS}
            for _ in range(2):  # the SAME path again: the yardstick for "equal gradients" (printed, not asserted on)
                again = finish(*run(name, None)[:3])
                assert torch.equal(again[0], ref[name][0]) and torch.equal(again[1], ref[name][1])
                for k in KEYS:
                    noise[name][k] = max(noise[name][k], rel_err(again[2][k], ref[name][2][k]))
        # K10 adds a Gaussian's tiles with float atomics in whatever order they finish; the deviation between two
        # IDENTICAL runs is heavy-tailed (a few ill-conditioned large splats): typically 1e-6, measured up to 2.3e-5 on
        # the large-splat scene.  The bar for the late path is north_star's 1e-4.
        print("run-to-run gradient deviation of the polled path:", noise)
        for name in scenes:
            noise[name] = {k: 1e-4 for k in KEYS}
        assert pairs["large"] > 2 * pairs["small"] > 0, pairs
        dgr.release_workspaces()
        dgr.set_speculative_sort(True)
        run("small", None)  # sizes the scratch: the capacity is now the small view's
        # (1) the op settles by itself: the large view outgrows the capacity
        gg, img, nc, n_render = run("large", None)
        assert n_render == pairs["large"]
        got = finish(gg, img, nc)
        assert torch.equal(got[0], ref["large"][0]) and torch.equal(got[1], ref["large"][1])
        for k in KEYS:  # (K10 adds with float atomics: the order of a Gaussian's tiles varies from run to run)
            assert rel_err(got[2][k], ref["large"][2][k]) <= noise["large"][k], k
        # (2) the caller collects: two views in flight before the first count is looked at; the first one overflows
        dgr.release_workspaces()
        run("small", None)
        late = []
        gl, il, ncl, _ = run("large", late)
        gs_, is_, ncs, _ = run("small", late)
        assert len(late) == 2
        counts = [settle() for settle in late]
        assert counts == [pairs["large"], pairs["small"]]
        for name, (gg, img, nc) in (("large", (gl, il, ncl)), ("small", (gs_, is_, ncs))):
            got = finish(gg, img, nc)
            assert torch.equal(got[0], ref[name][0]) and torch.equal(got[1], ref[name][1]), name
            for k in KEYS:
                assert rel_err(got[2][k], ref[name][2][k]) <= noise[name][k], (name, k)
    finally:
        dgr.set_speculative_sort(True)
        dgr.release_workspaces()


def _binning_views(device, W=333, H=211):
    from oracle import cref as C

    cam = S.orbit_cameras(4, W, H)[1]

    def inputs(N, sc, seed):
        g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
        m2, rgb, co, radii, depths, _, _ = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam))
        mask = _full_mask(cam)
        mask[1, :] = False
        return [t.to(device) for t in (m2, depths, radii, co, mask.view(-1).to(torch.uint8))]

    return [inputs(3000, 0.02, 5), inputs(9000, 0.03, 6), inputs(20000, 0.06, 7), inputs(9000, 0.05, 8)]


@pytest.mark.parametrize("which,grid_s", [("s", None), ("s", 5), ("p", None), ("ps", 3)])
def test_persistent_binning_survives_an_aborted_first_barrier(device, monkeypatch, which, grid_s):
    """Round 6 (verdict r05 item 4: no trap in the product path).  GSR_BIN_FORCE_ABORT makes the first grid barrier of the
    prepare ("p") and / or the sort ("s") kernel abort exactly as its time-out would: the prepare step repeats itself on
    the look-back pipeline (GSR_ERETRY inside the operator), in the sort step every workgroup but number 0 leaves and
    that one sorts the view alone inside the same launch.  Lists, ranges and counts must be those of the look-back
    pipeline, bit for bit, through the sized and the speculative (bounded) sort; the recovery counter moves; no fault"""
    import diff_gaussian_rasterization as dgr

    W, H = 333, 211
    views = _binning_views(device, W, H)
    dgr.release_workspaces()
    dgr.set_bin_persistent("off")
    try:
        ref = []
        for x in views:
            pl, rg, D = dgr.bin_gaussians(*x, W, H)
            ref.append((pl.clone(), rg.clone(), D))
        monkeypatch.setenv("GSR_BIN_PERSIST_MAXD", "1000000000")
        monkeypatch.setenv("GSR_BIN_FORCE_ABORT", which)
        if grid_s is not None:
            monkeypatch.setenv("GSR_BIN_GRID_S", str(grid_s))
        torch.cuda.synchronize()
        dgr.set_bin_rowmajor(False)  # (the persistent SORT kernel is what is under test)
        dgr.set_bin_persistent("both")  # (also ends the back-off an earlier recovery may have started)
        before = dgr.bin_persist_status()
        for spec in (False, True):
            dgr.release_workspaces()
            dgr.set_speculative_sort(spec)
            for v in (1, 1, 0, 2, 3, 1, 2):
                pl, rg, D = dgr.bin_gaussians(*views[v], W, H)
                rpl, rrg, rD = ref[v]
                assert D == rD, (spec, v)
                assert torch.equal(rg, rrg), (spec, v)
                assert pl.numel() == D and torch.equal(pl, rpl), (spec, v)
        torch.cuda.synchronize()
        after = dgr.bin_persist_status()
        assert after["faults"] == before["faults"]
        if "s" in which:
            assert after["solo_recoveries"] >= before["solo_recoveries"] + 14, (before, after)
        else:
            assert after["solo_recoveries"] == before["solo_recoveries"]
    finally:
        dgr.set_bin_persistent("env")
        dgr.set_bin_rowmajor("env")
        dgr.set_speculative_sort(True)
        dgr.release_workspaces()


def _shared_device_worker(rank, q, iters):
    try:
        import os
        import sys

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (os.path.join(root, "grendel-gs_amd"), root, os.path.join(root, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["GSR_BIN_PERSIST_MAXD"] = "1000000000"
        os.environ["GSR_BIN_SORT_TIMEOUT_MS"] = "20"
        os.environ["GSR_BIN_ROWS"] = "0"  # (the persistent SORT kernel is what is under test)
        import diff_gaussian_rasterization as dgr

        device = torch.device("cuda:0")
        torch.cuda.set_device(device)
        W, H = 1920, 1080
        g = S.make_gaussians(200_000, W, H, seed=11 + rank, scale_coef=0.004)
        cams = S.orbit_cameras(3, W, H)
        views = []
        for cam in cams:
            rast = dgr.GaussianRasterizer(settings_from(cam, torch.zeros(3)))
            gg = {k: v.to(device) for k, v in g.items()}
            m2, rgb, co, radii, depths = rast.preprocess_gaussians(*[gg[k] for k in KEYS], {})
            mask = torch.ones(((H + 15) // 16) * ((W + 15) // 16), dtype=torch.uint8, device=device)
            views.append([m2.detach(), depths.detach(), radii, co.detach(), mask])
        dgr.set_bin_persistent("off")
        ref = []
        for x in views:
            pl, rg, D = dgr.bin_gaussians(*x, W, H)
            ref.append((pl.clone(), rg.clone(), D))
        dgr.set_bin_persistent("both")
        bad = 0
        for i in range(iters):
            v = i % len(views)
            pl, rg, D = dgr.bin_gaussians(*views[v], W, H)
            rpl, rrg, rD = ref[v]
            if D != rD or not torch.equal(rg, rrg) or not torch.equal(pl, rpl):
                bad += 1
        torch.cuda.synchronize()
        q.put((rank, bad, dgr.bin_persist_status(), None))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, -1, None, traceback.format_exc() + repr(e)))


def test_two_processes_share_a_device_with_persistent_binning():
    """Two PROCESSES on one device, both running the persistent prepare AND sort kernels whatever the pair count (each
    asks for the whole device; their grids cannot both be resident): every view must finish with the look-back
    pipeline's lists -- through the prepare kernel's repeat on the look-back pipeline, the sort kernel's one-workgroup
    recovery and the back-off that follows -- and no process may die (ABI 11 trapped here after two seconds)"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_device_worker, args=(r, q, 150)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, bad, status, err in results:
        assert err is None, err
        assert bad == 0, (rank, bad, status)
        assert status["faults"] == 0, status
    print("persist status per process:", [r[2] for r in results])
    assert all(p.exitcode == 0 for p in procs)


def test_two_host_threads_on_two_streams_bin_the_same_lists(device):
    """K3-K7 called from TWO host threads, each on its own stream (bench.py's two-thread views leg; a render server with a
    thread per stream): the per-(device, stream) scratch, the count slots and the admission of the barrier kernels (the
    second stream's launch takes the look-back pipeline while the first one's persistent kernel is in flight) must give
    the lists of the sequential calls, bit for bit"""
    import threading

    import diff_gaussian_rasterization as dgr
    from oracle import cref as C

    W, H = 333, 211
    cam = S.orbit_cameras(4, W, H)[1]

    def inputs(N, sc, seed):
        g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
        m2, rgb, co, radii, depths, _, _ = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam))
        mask = _full_mask(cam)
        mask[1, :] = False
        return [t.to(device) for t in (m2, depths, radii, co, mask.view(-1).to(torch.uint8))]

    views = [inputs(3000, 0.02, 5), inputs(9000, 0.03, 6), inputs(20000, 0.06, 7), inputs(9000, 0.05, 8)]
    dgr.release_workspaces()
    ref = []
    for x in views:
        pl, rg, D = dgr.bin_gaussians(*x, W, H)
        ref.append((pl.clone(), rg.clone(), D))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
    got, errs = [[], []], []

    def worker(k):
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(streams[k]):
                for i in range(24):
                    v = (2 * i + k + i // 5) % len(views)
                    pl, rg, D = dgr.bin_gaussians(*views[v], W, H)
                    got[k].append((v, pl.clone(), rg.clone(), D))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    try:
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
    finally:
        dgr.release_workspaces()
    assert not errs, errs
    for k in range(2):
        assert len(got[k]) == 24
        for v, pl, rg, D in got[k]:
            rpl, rrg, rD = ref[v]
            assert D == rD, (k, v)
            assert torch.equal(rg, rrg), (k, v)
            assert pl.numel() == D and torch.equal(pl, rpl), (k, v)


@pytest.mark.parametrize("N,W,H,sc,seed,ci", SCENES[1:4])
def test_exact_tile_culling_changes_nothing_but_the_pair_count(device, N, W, H, sc, seed, ci):
    """gsr_set_tile_cull(1): fewer (tile, Gaussian) pairs, the same image and gradients -- against the run without culling
    (<= 2e-6 / 2e-5: the dropped pairs contribute nothing, only the blend's chunk boundaries move) and against the C oracle
    (the usual 1e-4), which knows no culling at all"""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer

    g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
    cam = S.orbit_cameras(4, W, H)[ci]
    bg = torch.tensor([0.2, 0.1, 0.4])
    mask = _full_mask(cam)
    wgt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(5))
    ref = oracle_c_chain(g, cam, bg, mask, wgt)
    res = {}
    try:
        for cull in (False, True):
            dgr.set_tile_cull(cull)
            rast = GaussianRasterizer(settings_from(cam, bg))
            gg = {k: v.to(device).requires_grad_(True) for k, v in g.items()}
            a, b, c, d, e = rast.preprocess_gaussians(*[gg[k] for k in KEYS], {})
            img, _, _, _ = rast.render_gaussians(a, c, b, e, d, mask.to(device), None, {"stats_collector": {}})
            (img * wgt.to(device)).sum().backward()
            res[cull] = (img.detach().cpu(), {k: gg[k].grad.detach().cpu() for k in KEYS},
                         dgr._RenderGaussians.last_num_rendered)
    finally:
        dgr.set_tile_cull("env")
    assert res[True][2] < res[False][2], (res[True][2], res[False][2])
    print(f"pairs {res[False][2]} -> {res[True][2]} ({res[True][2] / res[False][2]:.3f})")
    assert rel_err(res[True][0], res[False][0]) < 2e-6
    assert rel_err(res[True][0], ref["image"]) < 1e-4
    for k in KEYS:
        assert rel_err(res[True][1][k], res[False][1][k]) < 2e-5, k
        assert rel_err(res[True][1][k], ref["d_" + k]) < 1e-4, k


@pytest.mark.parametrize("N,W,H,sc,seed,ci", SCENES)
@pytest.mark.parametrize("bgv,sh_degree", [(0.0, 3), (0.7, 3), (0.0, 0), (0.7, 1), (0.0, 2)])
def test_full_chain_matches_c_oracle(device, N, W, H, sc, seed, ci, bgv, sh_degree):
    """preprocess -> render -> backward through the operator surface vs the C restatement: norm-wise AND p99
    element-wise on every gradient tensor, for every SH degree"""
    from diff_gaussian_rasterization import GaussianRasterizer

    g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc, sh_rest_sigma=0.1 if sh_degree == 3 else 0.4)
    cam = S.orbit_cameras(4, W, H)[ci]
    bg = torch.tensor([bgv, bgv * 0.5, 1.0 - bgv])
    mask = _full_mask(cam)
    if bgv > 0:
        mask[0:2, :] = False
    gen = torch.Generator().manual_seed(11)
    wgt = torch.rand(3, H, W, generator=gen)
    ref = oracle_c_chain(g, cam, bg, mask, wgt, sh_degree=sh_degree)

    rast = GaussianRasterizer(settings_from(cam, bg, sh_degree=sh_degree))
    gg = {k: v.to(device).requires_grad_(True) for k, v in g.items()}
    cuda_args = {"stats_collector": {}}
    m2, rgb, co, radii, depths = rast.preprocess_gaussians(gg["means3D"], gg["scales"], gg["rotations"], gg["shs"],
                                                           gg["opacities"], cuda_args)
    m2.retain_grad(); rgb.retain_grad(); co.retain_grad()
    img, n_render, _, n_contrib = rast.render_gaussians(means2D=m2, conic_opacity=co, rgb=rgb, depths=depths,
                                                        radii=radii, compute_locally=mask.to(device),
                                                        extended_compute_locally=None, cuda_args=cuda_args)
    assert img.shape == (3, H, W)
    # non-local tiles are exactly zero
    pm = mask.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    assert img[:, ~pm.to(device)].abs().sum().item() == 0.0
    assert rel_err(img, ref["image"]) < RTOL
    assert frac_bad(img, ref["image"], rtol=1e-3, atol=1e-4) < 2e-4
    assert n_contrib.shape == (H, W) and n_contrib.dtype == torch.int32  # positions in the (culled) HIP lists
    (img * wgt.to(device)).sum().backward()
    worst = 0.0
    for name, got, want in [("d_rgb", rgb.grad, ref["d_rgb"]), ("d_conic_opacity", co.grad, ref["d_conic_opacity"]),
                            ("d_means2D", m2.grad, ref["d_means2D"])] + \
            [(rk, gg[k].grad, ref[rk]) for k, rk in [("means3D", "d_means3D"), ("scales", "d_scales"),
                                                     ("rotations", "d_rotations"), ("shs", "d_shs"),
                                                     ("opacities", "d_opacities")]]:
        assert rel_err(got, want) < RTOL, name
        x = elem_excess(got, want)
        worst = max(worst, x)
        assert x <= 1.0, f"{name}: p99 element-wise excess {x:.2f} (|a-b| <= 1e-4 |b| + 1e-5 rms)"
    used = (sh_degree + 1) ** 2
    assert float(gg["shs"].grad[:, used:].abs().sum()) == 0.0, "coefficients above the active degree get no gradient"
    print(f"[full chain N={N} {W}x{H} deg={sh_degree}] worst p99 element-wise excess {worst:.2f}")
    st = cuda_args["stats_collector"]
    assert isinstance(st["forward_render_time"], float) and isinstance(st["backward_render_time"], float)


@pytest.mark.parametrize("sh_degree", [3, 1])
def test_small_chain_matches_fp64_autograd_oracle(device, sh_degree):
    """the arbiter: float64 autograd oracle (independent backward derivation)"""
    from diff_gaussian_rasterization import GaussianRasterizer
    from oracle import torch_oracle as O

    N, W, H = 600, 112, 80
    g = S.make_gaussians(N, W, H, seed=5, scale_coef=0.015)
    cam = S.orbit_cameras(4, W, H)[3]
    bg = torch.tensor([0.1, 0.4, 0.9])
    mask = _full_mask(cam)
    gen = torch.Generator().manual_seed(2)
    wgt = torch.rand(3, H, W, generator=gen)
    ins = {k: v.double().clone().requires_grad_(True) for k, v in g.items()}
    kw = cam_kwargs(cam, sh_degree)
    m2o, rgbo, coo, radiio, deptho = O.preprocess(*[ins[k] for k in KEYS], **kw)
    imgo, _, _ = O.render(m2o, coo, rgbo, deptho, radiio, mask, bg=bg, W=W, H=H)
    (imgo * wgt.double()).sum().backward()

    rast = GaussianRasterizer(settings_from(cam, bg, sh_degree=sh_degree))
    gg = {k: v.to(device).requires_grad_(True) for k, v in g.items()}
    m2, rgb, co, radii, depths = rast.preprocess_gaussians(*[gg[k] for k in KEYS], {})
    img, _, _, _ = rast.render_gaussians(m2, co, rgb, depths, radii, mask.to(device), None, {})
    (img * wgt.to(device)).sum().backward()
    assert rel_err(img, imgo) < RTOL
    for k in KEYS:
        assert rel_err(gg[k].grad, ins[k].grad) < RTOL, k
        assert elem_excess(gg[k].grad, ins[k].grad) <= 1.0, f"{k}: p99 element-wise"


def test_local2j_matches_oracle(device):
    from diff_gaussian_rasterization import _C
    from oracle import cref as C

    N, W, H = 4000, 320, 200
    g = S.make_gaussians(N, W, H, seed=9, scale_coef=0.02)
    cam = S.orbit_cameras(4, W, H)[0]
    m2, rgb, co, radii, depths, _, _ = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    for rows in ([0, 4, 9, gy], [0, gy], [0, 1, 2, 3, gy]):
        div = torch.tensor(rows, dtype=torch.int32) * gx
        ref = C.get_local2j_ids_bool(H, W, len(rows) - 1, m2, radii, div)
        out = _C.get_local2j_ids_bool(H, W, 0, len(rows) - 1, m2.to(device), radii.to(device), div.to(device), {})
        assert out.dtype == torch.bool and torch.equal(out.cpu(), ref)


def test_partition_union_equals_single(device):
    """W in {2,4} row-band partitions, rendered separately and SUMmed, reproduce the W=1 image
    exactly (tiles are independent; non-local pixels are 0) and the summed gradients match."""
    from diff_gaussian_rasterization import GaussianRasterizer

    N, W, H = 6000, 320, 208
    g = S.make_gaussians(N, W, H, seed=21, scale_coef=0.01)
    cam = S.orbit_cameras(4, W, H)[1]
    bg = torch.tensor([0.3, 0.3, 0.3])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    gen = torch.Generator().manual_seed(4)
    wgt = torch.rand(3, H, W, generator=gen).to(device)
    rast = GaussianRasterizer(settings_from(cam, bg))

    def run(bands):
        gg = {k: v.to(device).requires_grad_(True) for k, v in g.items()}
        m2, rgb, co, radii, depths = rast.preprocess_gaussians(*[gg[k] for k in KEYS], {})
        total = torch.zeros(3, H, W, device=device)
        for (l, r) in bands:
            mask = torch.zeros(gy, gx, dtype=torch.bool, device=device)
            mask[l:r] = True
            img, _, _, _ = rast.render_gaussians(m2, co, rgb, depths, radii, mask, None, {})
            total = total + img
        (total * wgt).sum().backward()
        return total.detach(), {k: gg[k].grad for k in KEYS}

    img1, gr1 = run([(0, gy)])
    for bands in ([(0, 5), (5, gy)], [(0, 3), (3, 6), (6, 10), (10, gy)]):
        imgw, grw = run(bands)
        assert torch.equal(imgw, img1), "forward composite is tile-independent: bitwise equal"
        for k in KEYS:
            assert rel_err(grw[k], gr1[k]) < RTOL, k
            assert elem_excess(grw[k], gr1[k]) <= 1.0, f"{k}: p99 element-wise (only the atomic-add order differs)"


def test_degenerate_inputs(device):
    from diff_gaussian_rasterization import GaussianRasterizer

    W, H = 64, 48
    cam = S.SyntheticCamera(0, W, H)
    bg = torch.tensor([0.2, 0.4, 0.6])
    rast = GaussianRasterizer(settings_from(cam, bg))
    mask = _full_mask(cam, device)
    # behind the camera, zero opacity, a single visible one
    means = torch.tensor([[0.0, 0.0, -3.0], [0.0, 0.0, 0.1], [0.1, 0.1, 4.0], [0.0, 0.0, 5.0]], device=device)
    scales = torch.full((4, 3), 0.05, device=device)
    rots = torch.tensor([[1.0, 0, 0, 0]] * 4, device=device)
    shs = torch.zeros(4, 16, 3, device=device); shs[:, 0] = 1.0
    opac = torch.tensor([[0.9], [0.9], [0.001], [0.8]], device=device)
    m2, rgb, co, radii, depths = rast.preprocess_gaussians(means, scales, rots, shs, opac, {})
    assert radii[0].item() == 0 and radii[1].item() == 0 and radii[2].item() > 0 and radii[3].item() > 0
    img, _, _, nc = rast.render_gaussians(m2, co, rgb, depths, radii, mask, None, {})
    assert torch.isfinite(img).all()
    # far corner sees only background
    assert torch.allclose(img[:, 0, 0].cpu(), bg, atol=1e-6)
    # empty set of Gaussians
    e = torch.zeros(0, 3, device=device)
    m2, rgb, co, radii, depths = rast.preprocess_gaussians(e, e, torch.zeros(0, 4, device=device),
                                                           torch.zeros(0, 16, 3, device=device),
                                                           torch.zeros(0, 1, device=device), {})
    img, _, _, _ = rast.render_gaussians(m2, co, rgb, depths, radii, mask, None, {})
    assert torch.allclose(img.cpu(), bg.view(3, 1, 1).expand(3, H, W))


def test_saturating_stack_early_stop(device):
    """many opaque splats on one pixel: T must stop above 1e-4 and n_contrib must match the oracle"""
    from diff_gaussian_rasterization import GaussianRasterizer
    from oracle import cref as C

    W, H = 32, 32
    cam = S.SyntheticCamera(0, W, H)
    n = 300
    g = dict(means3D=torch.cat([torch.zeros(n, 2), torch.linspace(2, 9, n)[:, None]], 1),
             scales=torch.full((n, 3), 0.3), rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1),
             shs=torch.rand(n, 16, 3, generator=torch.Generator().manual_seed(0)) * 0.5,
             opacities=torch.full((n, 1), 0.6))
    bg = torch.zeros(3)
    mask = _full_mask(cam)
    wgt = torch.ones(3, H, W)
    ref = oracle_c_chain(g, cam, bg, mask, wgt)
    rast = GaussianRasterizer(settings_from(cam, bg))
    gg = {k: v.to(device).requires_grad_(True) for k, v in g.items()}
    m2, rgb, co, radii, depths = rast.preprocess_gaussians(*[gg[k] for k in KEYS], {})
    img, _, _, nc = rast.render_gaussians(m2, co, rgb, depths, radii, mask.to(device), None, {})
    assert ref["final_T"].min().item() >= 1e-4 * 0.99
    assert int(nc.max()) <= n and int(nc.max()) < int(ref["n_contrib"].max()) + 1
    assert rel_err(img, ref["image"]) < RTOL
    img.sum().backward()
    # In this scene every pixel of the stack stops at the T < 1e-4 rule.  A pixel whose T lands within an ulp of 1e-4
    # may stop one entry earlier / later than in the restatement (v_exp_f32 vs libm expf): a discrete change of one
    # blended entry on that pixel, not rounding noise.  Measured (printed): <= 1e-4 norm-wise still holds.
    e = rel_err(gg["opacities"].grad, ref["d_opacities"])
    flips = int((nc.cpu() != ref["n_contrib"]).sum())
    print(f"[saturating] d_opacities rel {e:.2e}, n_contrib differs on {flips} of {W * H} pixels")
    assert e < RTOL
    for k, rk in [("means3D", "d_means3D"), ("scales", "d_scales"), ("shs", "d_shs")]:
        assert rel_err(gg[k].grad, ref[rk]) < RTOL, k


def test_batched_exchange_need_equals_per_camera_k2(device):
    """gsr_exchange_need (whole batch, destination-major) == K2 per camera (get_local2j_ids_bool) column by column"""
    g = torch.Generator().manual_seed(5)
    B, P, W, width, height = 3, 20000, 4, 640, 360
    gy = (height + 15) // 16
    m2 = (torch.rand(B, P, 2, generator=g) * torch.tensor([width * 1.2, height * 1.2]) - 20.0).to(device)
    radii = torch.randint(0, 60, (B, P), generator=g, dtype=torch.int32).to(device)
    bands = torch.zeros(B, W, 2, dtype=torch.int32)
    parts = [[0, 5, 11, gy], [0, gy], [0, 7, gy]]  # camera k is cut into len-1 bands ...
    owners = [[0, 1, 3], [2], [3, 0]]              # ... rendered by these global ranks
    for k in range(B):
        for j, r in enumerate(owners[k]):
            bands[k, r, 0], bands[k, r, 1] = parts[k][j], parts[k][j + 1]
    need, counts = dgr.exchange_need(m2, radii, bands, width, height)
    assert torch.equal(counts.cpu(), need.sum(2).int().cpu())
    for k in range(B):
        div = torch.tensor(parts[k], dtype=torch.int32, device=device) * ((width + 15) // 16)
        ref = dgr._C.get_local2j_ids_bool(height, width, 0, len(owners[k]), m2[k], radii[k], div, {})
        for r in range(W):
            want = ref[:, owners[k].index(r)] if r in owners[k] else torch.zeros(P, dtype=torch.bool, device=device)
            assert torch.equal(need[r, k], want), (k, r)


def test_fused_exchange_kernels_match_restatement(device):
    """gsr_exchange_count / gsr_exchange_pack / gsr_scatter_add_rows against their torch restatement
    (oracle/exchange_oracle.py): counts, per-chunk counts, message rows (bit-exact, incl. the radius bits), row order
    (destination, camera, local index) and the send index list; per-camera pack launches from one batch-wide count"""
    from oracle import exchange_oracle as XO

    g = torch.Generator().manual_seed(7)
    B, P, W, width, height = 3, 5000, 4, 640, 360
    gy = (height + 15) // 16
    m2 = (torch.rand(B, P, 2, generator=g) * torch.tensor([width * 1.2, height * 1.2]) - 20.0)
    radii = torch.randint(0, 60, (B, P), generator=g, dtype=torch.int32)
    rgb, co, depths = torch.rand(B, P, 3, generator=g), torch.rand(B, P, 4, generator=g), torch.rand(B, P, generator=g)
    bands = torch.zeros(B, W, 2, dtype=torch.int32)
    parts, owners = [[0, 5, 11, gy], [0, gy], [0, 7, gy]], [[0, 1, 3], [2], [3, 0]]
    for k in range(B):
        for j, r in enumerate(owners[k]):
            bands[k, r, 0], bands[k, r, 1] = parts[k][j], parts[k][j + 1]
    dm2, drad, drgb, dco, ddep, dbands = [t.to(device) for t in (m2, radii, rgb, co, depths, bands)]
    cc_ref, cnt_ref = XO.exchange_count(m2, radii, bands, 0, B, width, height)
    cc, cnt = dgr.exchange_count(dm2, drad, dbands, 0, B, width, height)
    assert torch.equal(cnt.cpu(), cnt_ref) and torch.equal(cc.cpu(), cc_ref)
    # the K2 mask path agrees
    need, counts = dgr.exchange_need(dm2, drad, dbands, width, height)
    assert torch.equal(counts.cpu(), cnt_ref)

    def seg(cams):
        off, o = [], 0
        for gdst in range(W):
            for k in cams:
                off.append(o)
                o += int(cnt_ref[gdst, k])
        return off, o

    for cams in ([0, 1, 2], [1], [2]):  # the whole batch in one launch / one camera per launch (pipelined exchanges)
        off, n_send = seg(cams)
        msg_ref, idx_ref = XO.exchange_pack(m2, rgb, co, radii, depths, bands, None, off, n_send, cams[0], len(cams), width,
                                            height)
        msg, idx = dgr.exchange_pack(dm2, drgb, dco, drad, ddep, dbands, cc, off, n_send, cams[0], len(cams), width,
                                     height, count_cameras=B, count_first=0)
        assert torch.equal(idx.cpu(), idx_ref), cams
        assert torch.equal(msg.cpu().view(torch.int32), msg_ref.view(torch.int32)), cams  # bit-exact records
    # scatter-add: duplicates accumulate
    n = 20000
    idx = torch.randint(0, 3000, (n,), generator=g, dtype=torch.int32)
    src = torch.randn(n, 9, generator=g)
    src[::7] = 0.0
    out = dgr.scatter_add_rows(idx.to(device), src.to(device), 3000)
    ref = XO.scatter_add_rows(idx, src.double(), 3000)
    assert rel_err(out, ref) < 1e-6
    pre = torch.ones(3000, 9, device=device)
    dgr.scatter_add_rows(idx.to(device), src.to(device), 3000, dst=pre)
    assert rel_err(pre - 1.0, ref) < 1e-5


def test_exchange_kernels_many_destinations_many_chunks(device):
    """the same kernels at 70 destinations (running positions in two registers, destinations taken eight at a time with
    a ragged last block) and 69 chunks (the prefix over the preceding chunks loops twice), exact and slab layouts, and
    the unpack of a slab that is neither a multiple of 256 rows nor 16-byte aligned"""
    from oracle import exchange_oracle as XO

    g = torch.Generator().manual_seed(13)
    B, P, W, width, height = 1, 70000, 70, 800, 2000
    gy = (height + 15) // 16
    m2 = (torch.rand(B, P, 2, generator=g) * torch.tensor([width * 1.1, height * 1.1]) - 20.0)
    radii = torch.randint(0, 40, (B, P), generator=g, dtype=torch.int32)
    rgb, co, depths = torch.rand(B, P, 3, generator=g), torch.rand(B, P, 4, generator=g), torch.rand(B, P, generator=g)
    cuts = sorted(set([0, gy] + torch.randperm(gy - 1, generator=g)[:W - 3].add(1).tolist()))  # W - 2 bands
    bands = torch.zeros(B, W, 2, dtype=torch.int32)
    for j in range(len(cuts) - 1):  # ranks 1 .. W - 2 render; ranks 0 and W - 1 render nothing of this camera
        bands[0, j + 1, 0], bands[0, j + 1, 1] = cuts[j], cuts[j + 1]
    dm2, drad, drgb, dco, ddep, dbands = [t.to(device) for t in (m2, radii, rgb, co, depths, bands)]
    cc_ref, cnt_ref = XO.exchange_count(m2, radii, bands, 0, B, width, height)
    cc, cnt = dgr.exchange_count(dm2, drad, dbands, 0, B, width, height)
    assert torch.equal(cnt.cpu(), cnt_ref) and torch.equal(cc.cpu(), cc_ref)
    assert int(cnt_ref[0, 0]) == 0 and int(cnt_ref[W - 1, 0]) == 0 and int(cnt_ref.sum()) > P
    off, o = [], 0
    for gdst in range(W):
        off.append(o)
        o += int(cnt_ref[gdst, 0])
    msg_ref, idx_ref = XO.exchange_pack(m2, rgb, co, radii, depths, bands, None, off, o, 0, 1, width, height)
    msg, idx = dgr.exchange_pack(dm2, drgb, dco, drad, ddep, dbands, cc, off, o, 0, 1, width, height)
    assert torch.equal(idx.cpu(), idx_ref)
    assert torch.equal(msg.cpu().view(torch.int32), msg_ref.view(torch.int32))
    caps = [[n + 64, max(n - 3, 0), n][gdst % 3] for gdst, n in enumerate(cnt_ref[:, 0].tolist())]
    msg_ref, idx_ref = XO.exchange_pack_slab(m2, rgb, co, radii, depths, bands, None, None, caps, 0, 1, width, height)
    msg, idx = dgr.exchange_pack_slab(dm2, drgb, dco, drad, ddep, dbands, cc, cnt, caps, 0, 1, width, height)
    assert torch.equal(idx.cpu(), idx_ref)
    assert torch.equal(msg.cpu().view(torch.int32), msg_ref.view(torch.int32))
    for first in (0, 1, 3):  # row offsets 0 / 44 / 132 bytes: the vector and the scalar staging path of the unpack
        part = msg[first:first + 1000 + first]
        outs = dgr.exchange_unpack(part)
        refs = XO.exchange_unpack(msg_ref[first:first + 1000 + first])
        for a, b in zip(outs, refs):
            assert torch.equal(a.cpu().reshape(b.shape), b), first


def test_slab_exchange_kernels_match_restatement(device):
    """gsr_exchange_pack_slab (capacity slabs: records at the front in the reference order, overflowing records dropped,
    zero padding with send index -1), gsr_exchange_unpack and the index -1 skip of gsr_scatter_add_rows against their
    torch restatements -- with capacities above, equal to and BELOW the true counts"""
    from oracle import exchange_oracle as XO

    g = torch.Generator().manual_seed(11)
    B, P, W, width, height = 2, 7000, 3, 640, 360
    gy = (height + 15) // 16
    m2 = (torch.rand(B, P, 2, generator=g) * torch.tensor([width * 1.2, height * 1.2]) - 20.0)
    radii = torch.randint(0, 60, (B, P), generator=g, dtype=torch.int32)
    rgb, co, depths = torch.rand(B, P, 3, generator=g), torch.rand(B, P, 4, generator=g), torch.rand(B, P, generator=g)
    bands = torch.zeros(B, W, 2, dtype=torch.int32)
    parts = [[0, 6, 14, gy], [0, 9, gy]]
    for k in range(B):
        for j in range(len(parts[k]) - 1):
            bands[k, j, 0], bands[k, j, 1] = parts[k][j], parts[k][j + 1]
    dm2, drad, drgb, dco, ddep, dbands = [t.to(device) for t in (m2, radii, rgb, co, depths, bands)]
    cc, cnt = dgr.exchange_count(dm2, drad, dbands, 0, B, width, height)
    true = cnt.cpu()
    for cams, rule in (([0], "above"), ([1], "above"), ([0], "equal"), ([1], "below"), ([0, 1], "mixed")):
        caps = []
        for gdst in range(W):
            for k in cams:
                n = int(true[gdst, k])
                caps.append({"above": (n * 5 // 4 + 256) // 256 * 256, "equal": n, "below": max(n - 37, 0),
                             "mixed": [n + 100, max(n - 5, 0), n][(gdst + k) % 3]}[rule])
        msg_ref, idx_ref = XO.exchange_pack_slab(m2, rgb, co, radii, depths, bands, None, None, caps, cams[0], len(cams),
                                                 width, height)
        msg, idx = dgr.exchange_pack_slab(dm2, drgb, dco, drad, ddep, dbands, cc, cnt, caps, cams[0], len(cams), width,
                                          height, count_cameras=B, count_first=0)
        assert torch.equal(idx.cpu(), idx_ref), (cams, rule)
        assert torch.equal(msg.cpu().view(torch.int32), msg_ref.view(torch.int32)), (cams, rule)
        outs = dgr.exchange_unpack(msg)
        refs = XO.exchange_unpack(msg_ref)
        for a, b in zip(outs, refs):
            assert torch.equal(a.cpu().reshape(b.shape), b), (cams, rule)
        assert int((outs[3] > 0).sum()) == sum(min(c, int(true[gdst, k])) for c, (gdst, k) in
                                                zip(caps, [(gd, k) for gd in range(W) for k in cams]))
    # padding rows (index -1) are skipped by the backward's scatter-add
    idx = torch.randint(-1, 500, (6000,), generator=g, dtype=torch.int32)
    src = torch.randn(6000, 9, generator=g)
    out = dgr.scatter_add_rows(idx.to(device), src.to(device), 500)
    assert rel_err(out, XO.scatter_add_rows(idx, src.double(), 500)) < 1e-6
    z = dgr.zeros_async((1000, 9), torch.float32, device)
    assert float(z.abs().sum()) == 0.0
