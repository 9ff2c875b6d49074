"""Write a tiny transforms-json dataset the reference's own loader reads.

scene/__init__.py:50-64 dispatches on the source path only: `sparse/` -> COLMAP binaries, "matrixcity" in the path ->
readCityInfo (scene/dataset_readers.py:255-350, 456-512: transforms_{train,test}.json with camera_angle_x + per-frame
c2w in OpenGL axes, `file_path` = image file incl. extension, and a tie-point .ply in the folder).  The NeRF-synthetic
reader exists but is never dispatched to, so the generated folder is named `matrixcity_*` and carries a random
points3d.ply (x y z nx ny nz red green blue, the layout storePly writes, :167-190).

The frames show a 3D-consistent scene (a few hundred coloured blobs inside the unit ball, splatted with numpy), so a
few dozen training iterations visibly lower the loss.  Test infrastructure for the `train.py runs unchanged` graft
test; there is no dataset on the GPU box.
"""
import json
import math
import os

import numpy as np


def _c2w(pos, target=np.zeros(3), up=np.array([0.0, 0.0, 1.0])):
    f = target - pos
    f = f / np.linalg.norm(f)
    z = -f  # OpenGL cameras look down -z
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, pos
    return m


def _render(c2w, fovx, W, H, centers, colors, sigma):
    m = c2w.copy()
    m[:3, 1:3] *= -1  # to COLMAP axes, as the loader does
    w2c = np.linalg.inv(m)
    pc = centers @ w2c[:3, :3].T + w2c[:3, 3]
    fx = W / (2.0 * math.tan(fovx / 2.0))
    z = np.maximum(pc[:, 2], 1e-3)
    u, v = fx * pc[:, 0] / z + W / 2.0, fx * pc[:, 1] / z + H / 2.0
    s = fx * sigma / z
    ys, xs = np.mgrid[0:H, 0:W]
    acc = np.zeros((H, W, 3))
    wsum = np.zeros((H, W))
    for i in np.argsort(-z):
        if pc[i, 2] < 0.2:
            continue
        w = np.exp(-((xs - u[i]) ** 2 + (ys - v[i]) ** 2) / (2.0 * s[i] ** 2))
        acc += w[..., None] * colors[i]
        wsum += w
    alpha = 1.0 - np.exp(-wsum)
    rgb = acc / np.maximum(wsum, 1e-6)[..., None]
    return np.concatenate([rgb * alpha[..., None], alpha[..., None]], axis=-1)


def _write_ply(path, xyz, rgb_u8):
    dt = [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("red", "u1"),
          ("green", "u1"), ("blue", "u1")]
    arr = np.zeros(xyz.shape[0], dtype=dt)
    arr["x"], arr["y"], arr["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    arr["red"], arr["green"], arr["blue"] = rgb_u8[:, 0], rgb_u8[:, 1], rgb_u8[:, 2]
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {len(arr)}"]
    head += [f"property {'uchar' if t == 'u1' else 'float'} {n}" for n, t in dt] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        f.write(arr.tobytes())


def generate(path, n_views=12, width=208, height=144, n_blobs=160, seed=0, fovx=0.9, n_points=20000):
    from PIL import Image

    assert "matrixcity" in path, "the reference's Scene only dispatches to the transforms reader for such paths"

    rs = np.random.RandomState(seed)
    d = rs.randn(n_blobs, 3)
    centers = d / np.linalg.norm(d, axis=1, keepdims=True) * rs.rand(n_blobs, 1) ** (1 / 3) * 0.9
    colors = rs.rand(n_blobs, 3)
    sigma = 0.04 + 0.05 * rs.rand(n_blobs)
    os.makedirs(os.path.join(path, "train"), exist_ok=True)
    os.makedirs(os.path.join(path, "test"), exist_ok=True)
    for split, n, phase in (("train", n_views, 0.0), ("test", max(2, n_views // 4), 0.37)):
        frames = []
        for k in range(n):
            th = 2.0 * math.pi * (k + phase) / n
            pos = np.array([3.2 * math.cos(th), 3.2 * math.sin(th), 0.8 + 0.6 * math.sin(2 * th)])
            c2w = _c2w(pos)
            img = _render(c2w, fovx, width, height, centers, colors, sigma)
            Image.fromarray((np.clip(img, 0, 1) * 255).astype(np.uint8), "RGBA").save(
                os.path.join(path, split, f"r_{k}.png"))
            frames.append({"file_path": os.path.join(os.path.abspath(path), split, f"r_{k}.png"),
                           "transform_matrix": c2w.tolist()})
        with open(os.path.join(path, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": fovx, "frames": frames}, f)
    _write_ply(os.path.join(path, "points3d.ply"), rs.rand(n_points, 3) * 2.2 - 1.1,
               (rs.rand(n_points, 3) * 255).astype(np.uint8))
    return path


if __name__ == "__main__":
    import sys

    print(generate(sys.argv[1] if len(sys.argv) > 1 else "/tmp/matrixcity_scene"))
