"""The closed form of the exact tile culling (csrc/binning_persist.h: gsr_tile_mask), restated in numpy float32 and held
against brute force in float64: per tile row of a Gaussian's rect, the span of tiles whose 16 x 16 block of pixel centres
the alpha >= 1/255 ellipse reaches.  Checked on random conics: (a) a tile the mask drops has alpha < 1/255 on EVERY pixel
centre of the tile (what the reference's per-pixel rule needs for the image to be unchanged); (b) the mask keeps little
more than the exact box test keeps.  The HIP code itself is held to (a) on the GPU
(tests/test_gpu_parity.py::test_binning_is_ordered_subsequence_of_reference_lists with culling on)."""
import numpy as np

F = np.float32


def tile_mask(cx, cy, A, B, C, o, minx, miny, maxx, maxy):
    """float32 restatement of gsr_tile_mask: -> set of kept (x, y) tiles of the rect"""
    w, h = maxx - minx, maxy - miny
    keep = set()
    det = F(A * C - B * B)
    if not (A > 0 and C > 0 and det > F(1e-6) * A * C) or not (o >= F(1.0 / 255.0)):
        return {(x, y) for x in range(minx, maxx) for y in range(miny, maxy)}
    lim = F(F(2.0 * 0.6931471805599453) * F(np.log2(F(255.0) * o)) * F(1.002) + F(0.01))
    rA = F(1.0) / A
    dxs = F(np.sqrt(lim * C / det))
    dys = F(B * dxs / C)
    for r in range(h):
        t0 = F((miny + r) * 16) - cy
        t1 = F(t0 + F(15.0))
        if -dys >= t0 and -dys <= t1:
            dxR, emptyR = dxs, False
        else:
            t = t0 if -dys < t0 else t1
            disc = F(lim * A - det * t * t)
            emptyR = disc < 0
            dxR = F((-B * t + F(np.sqrt(max(disc, F(0))))) * rA)
        if dys >= t0 and dys <= t1:
            dxL, emptyL = F(-dxs), False
        else:
            t = t0 if dys < t0 else t1
            disc = F(lim * A - det * t * t)
            emptyL = disc < 0
            dxL = F((-B * t - F(np.sqrt(max(disc, F(0))))) * rA)
        xa, xb = 0, w
        if emptyR and emptyL:
            continue
        if not emptyR and not emptyL:
            dxR = F(dxR + F(0.02) + F(1e-4) * abs(dxR))
            dxL = F(dxL - (F(0.02) + F(1e-4) * abs(dxL)))
            xa = max(0, int(np.ceil(F((cx + dxL - F(15.0)) * F(1.0 / 16.0)))) - minx)
            xb = min(w, int(np.floor(F((cx + dxR) * F(1.0 / 16.0)))) + 1 - minx)
            if xb <= xa:
                continue
        for c in range(xa, xb):
            keep.add((minx + c, miny + r))
    return keep


def box_min_q(cx, cy, A, B, C, x0, y0, x1, y1):
    """float64: minimum of A dx^2 + 2 B dx dy + C dy^2 over the box of pixel centres [x0, x1] x [y0, y1] (dense sampling
    of the continuous box's edges + the centre test: the form is convex)"""
    if x0 <= cx <= x1 and y0 <= cy <= y1:
        return 0.0
    best = np.inf
    ts = np.linspace(0.0, 1.0, 257)
    for (ax, ay, bx, by) in ((x0, y0, x1, y0), (x0, y1, x1, y1), (x0, y0, x0, y1), (x1, y0, x1, y1)):
        dx = ax + (bx - ax) * ts - cx
        dy = ay + (by - ay) * ts - cy
        best = min(best, float((A * dx * dx + 2 * B * dx * dy + C * dy * dy).min()))
    return best


def test_closed_form_spans_drop_only_what_contributes_nothing():
    rng = np.random.default_rng(7)
    dropped = kept = exact_kept = 0
    for it in range(600):
        # a random positive-definite conic (inverse covariance) with a screen footprint of 1 .. 6 tiles, any orientation
        s1, s2 = rng.uniform(2.0, 40.0), rng.uniform(2.0, 40.0)
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        cov = R @ np.diag([s1 * s1, s2 * s2]) @ R.T + 0.3 * np.eye(2)
        con = np.linalg.inv(cov)
        A, B, C = F(con[0, 0]), F(con[0, 1]), F(con[1, 1])
        o = F(rng.choice([0.004, 0.01, 0.05, 0.3, 0.9, 0.99]))
        cx, cy = F(rng.uniform(100, 400)), F(rng.uniform(100, 400))
        if o < 1.0 / 255.0:
            continue
        # the alpha-extent box (gsr_alpha_extent), as K3 builds the rect
        tau = np.log(255.0 * float(o))
        s = 2.0 * tau / float(A * C - B * B)
        ex, ey = np.sqrt(s * float(C)) * 1.01 + 0.5, np.sqrt(s * float(A)) * 1.01 + 0.5
        minx = int(np.ceil((float(cx) - ex - 15) / 16.0))
        maxx = int(np.floor((float(cx) + ex) / 16.0)) + 1
        miny = int(np.ceil((float(cy) - ey - 15) / 16.0))
        maxy = int(np.floor((float(cy) + ey) / 16.0)) + 1
        if maxx <= minx or maxy <= miny or (maxx - minx) * (maxy - miny) > 64:
            continue
        keep = tile_mask(cx, cy, A, B, C, o, minx, miny, maxx, maxy)
        lim_exact = 2.0 * tau
        Ad, Bd, Cd, od = float(A), float(B), float(C), float(o)
        for x in range(minx, maxx):
            for y in range(miny, maxy):
                q = box_min_q(float(cx), float(cy), Ad, Bd, Cd, 16.0 * x, 16.0 * y, 16.0 * x + 15, 16.0 * y + 15)
                exact_kept += q <= lim_exact
                if (x, y) in keep:
                    kept += 1
                    continue
                dropped += 1
                # (a) no pixel centre of a dropped tile reaches alpha >= 1/255
                px, py = np.meshgrid(np.arange(16 * x, 16 * x + 16), np.arange(16 * y, 16 * y + 16))
                dx, dy = px - float(cx), py - float(cy)
                power = -0.5 * (Ad * dx * dx + Cd * dy * dy) - Bd * dx * dy
                alpha = od * np.exp(power)
                assert alpha.max() < 1.0 / 255.0, (it, x, y, alpha.max() * 255)
                assert q > lim_exact, (it, x, y, q, lim_exact)
    assert dropped > 300, dropped              # the test does exercise culling ...
    assert kept <= 1.03 * exact_kept + 5, (kept, exact_kept)   # ... and the tolerance keeps few tiles the exact test drops
