"""Level-B1 graft: with grendel-gs_amd/b1_graft on the module path, the REFERENCE's own
Python -- gaussian_renderer, arguments, scene, train_internal -- imports unchanged against this repo's
operator module and shims.  Needs the reference tree (present in the build container only) -> skipped on
the GPU box.  Plus unit tests of the two off-path shims."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRAFT = os.path.join(ROOT, "grendel-gs_amd", "b1_graft")
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussian_renderer")), reason="reference tree not present")
def test_reference_python_imports_against_the_graft():
    code = (
        "import diff_gaussian_rasterization as d, gaussian_renderer as g, gaussian_renderer.workload_division as w, "
        "gaussian_renderer.loss_distribution as l, arguments, scene, train_internal\n"
        "assert d.__file__.startswith(%r), d.__file__\n"
        "assert g.__file__.startswith(%r) and w.__file__.startswith(%r)\n"
        "assert d._C.get_block_XY() == (16, 16, 256)\n"
        "import utils.general_utils as u\n"
        "u.set_block_size(*d._C.get_block_XY()); u.set_img_size(1080, 1920)\n"
        "assert (u.TILE_Y, u.TILE_X) == (68, 120)\n"
        "for n in ('distributed_preprocess3dgs_and_all2all_final','render_final'): assert hasattr(g, n)\n"
        "print('ok')\n" % (GRAFT, REF, REF))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REF, GRAFT]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def _graft_import(name):
    sys.path.insert(0, GRAFT)
    try:
        mod = __import__(name, fromlist=["x"])
    finally:
        sys.path.remove(GRAFT)
    return mod


def test_plyfile_shim_roundtrip(tmp_path):
    ply = _graft_import("plyfile")
    dt = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("f_dc_0", "f4")]
    arr = np.zeros(17, dtype=dt)
    rng = np.random.RandomState(0)
    for n, _ in dt:
        arr[n] = rng.rand(17) * 200
    el = ply.PlyElement.describe(arr, "vertex")
    p = str(tmp_path / "m.ply")
    ply.PlyData([el]).write(p)
    back = ply.PlyData.read(p)
    v = back.elements[0]
    assert [q.name for q in v.properties] == [n for n, _ in dt]
    for n, _ in dt:
        assert np.array_equal(np.asarray(back["vertex"][n]), arr[n])
    # ascii flavour
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty uchar red\nend_header\n"
                "1.5 3\n-2.0 250\n")
    a = ply.PlyData.read(p)["vertex"]
    assert list(a["x"]) == [1.5, -2.0] and list(a["red"]) == [3, 250]


def test_distcuda2_is_the_hip_operator_and_has_no_cpu_path():
    """simple_knn._C.distCUDA2 resolves to the library's K-NN operator; host tensors are refused loudly"""
    knn = _graft_import("simple_knn._C")
    pts = torch.rand(50, 3, generator=torch.Generator().manual_seed(0))
    with pytest.raises(RuntimeError, match="no CPU"):
        knn.distCUDA2(pts)


def test_knn_oracle_matches_kdtree():
    """pin the brute-force restatement (oracle/gsraster_ref.c) against an independent exact k-NN"""
    from scipy.spatial import cKDTree

    from oracle import cref as C

    g = torch.Generator().manual_seed(1)
    pts = torch.cat([torch.rand(3000, 3, generator=g), 0.01 * torch.randn(2000, 3, generator=g) + 0.5,
                     torch.rand(5, 3, generator=g).repeat(3, 1)])  # uniform + a dense cluster + exact duplicates
    out = C.knn_mean_dist2(pts)
    d, _ = cKDTree(pts.double().numpy()).query(pts.double().numpy(), k=4)
    ref = (d[:, 1:] ** 2).mean(1)
    assert np.allclose(out.double().numpy(), ref, rtol=2e-4, atol=1e-9)
    # degenerate sizes: mean over the neighbours that exist
    assert C.knn_mean_dist2(torch.zeros(1, 3)).tolist() == [0.0]
    two = C.knn_mean_dist2(torch.tensor([[0.0, 0, 0], [1.0, 0, 0]]))
    assert two.tolist() == [1.0, 1.0]
