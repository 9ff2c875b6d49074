"""Seeded input scenes for the level-B1 graft runs (tests/refgraft/run_iteration.py).  Test infrastructure.

numpy's legacy RandomState (MT19937) is bit-reproducible across hosts, so the same (name, seed) gives the same
inputs in the build container and on the GPU box; frozen fixtures nevertheless STORE their inputs.
"""
import math

import numpy as np


def _orbit(n_views, centroid_z=6.0):
    """R (camera class convention: R = Rw2c^T), T of n cameras rotated about (0, 0, centroid_z) around the y axis"""
    Rs, Ts = [], []
    c = np.array([0.0, 0.0, centroid_z])
    for k in range(n_views):
        th = 2.0 * math.pi * k / n_views * 0.25  # a quarter orbit: every view keeps most of the cloud in front
        Rw2c = np.array([[math.cos(th), 0.0, -math.sin(th)], [0.0, 1.0, 0.0], [math.sin(th), 0.0, math.cos(th)]])
        Rs.append(Rw2c.T.copy())
        Ts.append(c - Rw2c @ c)
    return np.stack(Rs), np.stack(Ts)


def make_scene(n, width, height, n_views, seed=0, scale_coef=0.01, tile_rows_heuristic=None):
    """raw GaussianModel parameters (scene/gaussian_model.py:219-242 layout), cameras (R, T, FoV) and uint8 ground
    truth, SURVEY.md 8(d) distributions"""
    rs = np.random.RandomState(seed)
    fx = 0.9 * width
    tanx, tany = width / (2.0 * fx), height / (2.0 * fx)
    z = rs.rand(n) * 8.0 + 2.0
    x = (rs.rand(n) * 2.3 - 1.15) * z * tanx
    y = (rs.rand(n) * 2.3 - 1.15) * z * tany
    xyz = np.stack([x, y, z], 1)
    scaling = np.log(scale_coef * z)[:, None] + 0.5 * rs.randn(n, 3)
    rotation = rs.randn(n, 4)  # NOT normalised: the getters' F.normalize is part of what is compared
    opacity = np.clip(2.0 * rs.randn(n, 1), -13.8, 13.8)
    f_dc = (rs.rand(n, 1, 3) * 2.0 - 1.0) / 0.28209479177387814
    f_rest = 0.1 * rs.randn(n, 15, 3)
    R, T = _orbit(n_views)
    gt = rs.randint(0, 256, size=(n_views, 3, height, width)).astype(np.uint8)
    out = dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity, features_dc=f_dc, features_rest=f_rest,
               cam_R=R, cam_T=T, cam_fovx=np.full(n_views, 2.0 * math.atan(width / (2.0 * fx))),
               cam_fovy=np.full(n_views, 2.0 * math.atan(height / (2.0 * fx))), gt=gt,
               width=np.int64(width), height=np.int64(height), bg=np.array([0.1, 0.2, 0.3], np.float32))
    for k in ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest"):
        out[k] = out[k].astype(np.float32)
    if tile_rows_heuristic is not None:
        out["heuristic"] = np.asarray(tile_rows_heuristic, np.float32)
    return out


def skewed_heuristic(n_views, tile_y, seed=3):
    """per-row costs that put the cut points away from the middle (and differ per camera)"""
    rs = np.random.RandomState(seed)
    h = 0.25 + rs.rand(n_views, tile_y)
    h[:, : tile_y // 3] *= 3.0
    return h.astype(np.float32)


CASES = {
    # name: (N, width, height, views in the dataset, bsz, world, iterations, fake timings, skewed heuristics)
    "c0": dict(n=3072, width=208, height=144, views=2, bsz=2, world=1, iters=1),
    "w2": dict(n=3072, width=208, height=144, views=2, bsz=1, world=2, iters=2, skew=True),
    "w2b2": dict(n=4096, width=256, height=256, views=2, bsz=2, world=2, iters=1),
    "w4b2": dict(n=4096, width=256, height=256, views=2, bsz=2, world=4, iters=1, skew=True),
    "w8b4": dict(n=8192, width=256, height=272, views=4, bsz=4, world=8, iters=2, skew=True),  # configs[3]'s shape
    "hd": dict(n=200_000, width=1920, height=1080, views=2, bsz=1, world=1, iters=1, scale_coef=0.004),
    "hdw2": dict(n=100_000, width=1920, height=1088, views=2, bsz=1, world=2, iters=4, fake_times=True,
                 scale_coef=0.004),
}


def build_case(name):
    c = dict(CASES[name])
    tile_y = (c["height"] + 15) // 16
    heur = skewed_heuristic(c["views"], tile_y) if c.get("skew") else None
    s = make_scene(c["n"], c["width"], c["height"], c["views"], seed=7, scale_coef=c.get("scale_coef", 0.01),
                   tile_rows_heuristic=heur)
    s.update(bsz=np.int64(c["bsz"]), world=np.int64(c["world"]), iters=np.int64(c["iters"]),
             fake_times=np.int64(1 if c.get("fake_times") else 0))
    return s


def fuzz_heuristics(seed, rows):
    """seeded per-row cost vectors of the cut-point fuzz (tests/refgraft/cuts_fuzz.py, test_gpu_reference_b1.py)"""
    rs = np.random.RandomState(seed)
    kind = seed % 4
    if kind == 0:    # measured-time-like: per band a constant time / rows
        h = np.repeat(rs.rand(8) * 3 + 0.05, (rows + 7) // 8)[:rows]
    elif kind == 1:  # smooth
        h = 0.2 + rs.rand(rows)
    elif kind == 2:  # spiky
        h = 0.01 + rs.rand(rows) ** 6 * 10
    else:            # the initial all-ones
        h = np.ones(rows)
    return h.astype(np.float32)


def fuzz_case(seed):
    return [68, 135, 68 * 4, 35 * 4][seed % 4], [2, 4, 8, 8][(seed // 4) % 4]
