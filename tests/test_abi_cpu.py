"""CPU checks of the drop-in boundary: the C-ABI library loads, exports exactly what include/gsraster.h
declares, and the operator mirror has the reference's surface and error behaviour (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gsraster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from diff_gaussian_rasterization import _lib

    names = _declared()
    assert len(names) >= 12
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/gsraster.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_pure_host_entry_points():
    import diff_gaussian_rasterization as dgr

    assert dgr._C.get_block_XY() == (16, 16, 256)  # utils/general_utils.py:78-79 expects these
    assert dgr._lib.lib.gsr_error_string(0) == b"success"
    assert b"invalid" in dgr._lib.lib.gsr_error_string(-1)
    assert dgr._lib.lib.gsr_bin_prepare_bytes(1000, 1920, 1080) > 0
    assert dgr._lib.lib.gsr_bin_sort_bytes(1000, 5000, 1920, 1080) > 5000 * 16
    lib = dgr._lib.lib
    assert b"barrier" in lib.gsr_error_string(-4)  # GSR_EFAULT (ABI 12: no kernel traps; a late barrier fault is reported)
    # workspace arithmetic of the sort (pure host code): the scratch for D pairs holds both layouts (two-pass pipelines:
    # 16 B / pair; row-major pipeline: 12 B / segment with R <= D, per-tile tables, look-back words), grows with D, and
    # gsr_bin_sort_capacity inverts it
    prev = 0
    for D in (1, 4095, 4096, 1 << 20, (5 << 20) - 1, 5 << 20, 14_000_000, 100_000_000):
        nbytes = lib.gsr_bin_sort_bytes(1000, D, 1920, 1080)
        assert nbytes >= 16 * D and nbytes >= prev, (D, nbytes)
        cap = lib.gsr_bin_sort_capacity(1000, nbytes, 1920, 1080)
        assert cap >= D and lib.gsr_bin_sort_bytes(1000, cap, 1920, 1080) <= nbytes, (D, cap)
        prev = nbytes
    assert lib.gsr_bin_sort_capacity(1000, 1 << 30, 8000, 8000) == 0  # > 256 x 256 tiles: no bounded sort
    # the device words a caller may read from the prepare workspace lie inside it
    for P in (1, 1000, 1_000_000):
        total = lib.gsr_bin_prepare_bytes(P, 1920, 1080)
        assert 0 < lib.gsr_bin_total_offset(P, 1920, 1080) < total
        assert 0 < lib.gsr_bin_segments_offset(P, 1920, 1080) < total
        assert lib.gsr_bin_total_offset(P, 1920, 1080) != lib.gsr_bin_segments_offset(P, 1920, 1080)
    for mode in (-1, 0, 1):
        assert lib.gsr_set_bin_rowmajor(mode) == 0
    assert lib.gsr_set_bin_rowmajor(2) == -1 and lib.gsr_set_bin_rowmajor(-1) == 0
    # argument validation happens before any device work
    assert lib.gsr_densify_stats(-1, None, None, 9, None, None, None, None) == -1
    assert lib.gsr_densify_stats(10, None, None, 1, None, None, None, None) == -1  # a row holds at least (x, y)
    assert lib.gsr_densify_stats(0, None, None, 9, None, None, None, None) == 0
    # ABI 13 (row bands as device data): a launch without its band words, or without a capacity, is refused
    assert lib.gsr_abi_version() == 13
    assert lib.gsr_l1_ssim_forward_band(3, 64, 64, None, 0, None, None, None, None, None, None, None) == -1
    assert lib.gsr_l1_ssim_backward_band(3, 0, 64, None, 0, None, None, None, None, None, None, 1.0, 1.0, None, 0, None,
                                         None) == -1
    assert lib.gsr_band_mask(40, 23, 1, None, 4, None, None) == -1
    assert lib.gsr_band_mask(40, 23, 0, None, 4, None, None) == -1
    rc = dgr._lib.lib.gsr_preprocess_forward(-1, 3, 16, *([None] * 2), 1.0, *([None] * 6), 10, 10, 1.0, 1.0,
                                             *([None] * 8))
    assert rc == -1


def test_operator_surface_matches_reference_names():
    import diff_gaussian_rasterization as dgr

    fields = dgr.GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                      "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    rs = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3,
                                           torch.zeros(3), False, False)
    r = dgr.GaussianRasterizer(raster_settings=rs)
    assert r.raster_settings.image_height == 8
    for name in ("preprocess_gaussians", "render_gaussians"):
        assert callable(getattr(r, name))
    for name in ("get_block_XY", "get_local2j_ids_bool", "get_local2j_ids_bool_adjust_mode6", "get_touched_locally",
                 "get_pixels_compute_locally_and_in_rect"):
        assert callable(getattr(dgr._C, name))
    with pytest.raises(NotImplementedError):
        dgr.load_image_tiles_by_pos()
    with pytest.raises(NotImplementedError):
        dgr._C.get_touched_locally(None, 1, 1, 0)


def test_no_cpu_fallback():
    """the product path must fail loudly on host tensors instead of silently computing elsewhere"""
    import diff_gaussian_rasterization as dgr

    rs = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3,
                                           torch.zeros(3), False, False)
    r = dgr.GaussianRasterizer(rs)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.preprocess_gaussians(torch.zeros(4, 3), torch.ones(4, 3), torch.ones(4, 4), torch.zeros(4, 16, 3),
                               torch.ones(4, 1), {})


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "grendel-gs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f"{f} imports oracle/"


def test_band_aware_xcd_map_is_a_balanced_bijection():
    """Python restatement of gsr_xcd_span_of_block_band (grendel-gs_amd/csrc/common.h): K8 / K10 hand every XCD a
    contiguous 1/8 of the BAND a rank renders.  Bijective over all tiles, every XCD gets |band| / 8 (+-1) band tiles as
    one contiguous run, and it equals the plain span map when the band is the whole grid.  (The device function itself is
    exercised by the GPU band-union test: the SUM of band renders is bitwise the full render.)"""
    import random
    import re

    src = open(os.path.join(ROOT, "grendel-gs_amd", "csrc", "common.h")).read()
    body = src[src.index("gsr_xcd_span_of_block_band"):]
    for needle in ("(hull - xcd + 7) >> 3", "xcd * (hull >> 3) + min(xcd, hull & 7)", "xcd * (nwg >> 3) + min(xcd, nwg & 7)",
                   "n < first ? n : n + hull"):
        assert needle in body, f"common.h no longer matches the restatement below: {needle}"
    assert re.search(r"if \(j < hl\) return first \+ hs \+ j;", body)

    def band_map(b, nwg, first, hull):
        xcd, j = b & 7, b >> 3
        hl = (hull - xcd + 7) >> 3
        hs = xcd * (hull >> 3) + min(xcd, hull & 7)
        if j < hl:
            return first + hs + j
        ts = xcd * (nwg >> 3) + min(xcd, nwg & 7)
        n = (ts - hs) + (j - hl)
        return n if n < first else n + hull

    def span(b, nwg):
        xcd, q, r = b & 7, nwg >> 3, nwg & 7
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (b >> 3)

    rnd = random.Random(0)
    cases = [(120, 68, 0, 68), (120, 68, 34, 43), (240, 135, 17, 34), (13, 9, 8, 9)]
    cases += [(gx, gy, lo, rnd.randint(lo + 1, gy)) for gx, gy, lo in
              ((rnd.randint(1, 40), g, rnd.randint(0, g - 1)) for g in (rnd.randint(1, 40) for _ in range(300)))]
    for gx, gy, lo, hi in cases:
        T, first, hull = gx * gy, lo * gx, (hi - lo) * gx
        m = [band_map(b, T, first, hull) for b in range(T)]
        assert sorted(m) == list(range(T)), (gx, gy, lo, hi)
        for x in range(8):
            bt = [t for b, t in enumerate(m) if b % 8 == x and first <= t < first + hull]
            assert not bt or bt == list(range(bt[0], bt[0] + len(bt)))
            assert abs(len(bt) - hull / 8) < 1 + 1e-9
        if hull == T:
            assert m == [span(b, T) for b in range(T)]


def test_bench_accounting_knows_the_fused_step():
    """bench.py's algorithmic bytes of the fused K11 + Adam launch = K11's + Adam's minus the gradient round trip and
    the optimizer's parameter read; tools/pmc_collect.py files the fused kernels under their own group"""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for B in (1, 4):
        m = {"N": 1000, "B": B, "M": 16}
        k11 = bench.algorithmic_bytes("preprocess_backward", m)
        adam = bench.algorithmic_bytes("adam", {"numel": 1000 * 59})
        fused = bench.algorithmic_bytes("preprocess_backward_adam", m)
        assert fused == k11 + adam - 1000 * 3 * 236 == 1000 * (1416 + 80 * B)
    spec = importlib.util.spec_from_file_location("pmc_collect", os.path.join(root, "tools", "pmc_collect.py"))
    pc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pc)
    assert pc.group_of("void (anonymous namespace)::preprocess_backward_adam_kernel<3>(int, float*)") == \
        "preprocess_backward_adam"
    assert pc.group_of("void (anonymous namespace)::preprocess_backward_adam_batched_kernel<3>(int, int)") == \
        "preprocess_backward_adam"
    assert pc.group_of("void (anonymous namespace)::preprocess_backward_kernel<3, true>(int, int)") == \
        "preprocess_backward"
    assert pc.group_of("(anonymous namespace)::adam_multi_kernel((anonymous namespace)::AdamBatch, float)") == "adam"


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus N` without a launcher becomes `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...` (the reference's start-up, README.md:199-202)"""
    import importlib.util
    import sys

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = bench.self_launch_command(4, ["--gpus", "4", "--steps", "7"])
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "7"]
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"  # set before torch is imported (graph replays)
