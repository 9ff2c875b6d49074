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


# ------------------------------------------------------------------------------- level B2
MIRROR = os.path.join(ROOT, "grendel-gs_amd", "gaussian_renderer")
MIRROR_FILES = ["__init__.py", "workload_division.py", "loss_distribution.py"]


def _utils_names_used_by_the_mirror():
    import ast

    names = set()
    for f in MIRROR_FILES:
        tree = ast.parse(open(os.path.join(MIRROR, f)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "utils":
                names.add(node.attr)
    names.discard("general_utils")
    return sorted(names)


def test_mirror_only_uses_names_the_reference_utils_module_has():
    """graft level B2 keeps the reference's utils/: every `utils.X` the three mirror files touch must exist there.
    The list of the reference's names travels as a golden (tests/golden/reference_utils_names.txt, written by
    tests/golden/make_golden.py from the reference's utils/general_utils.py) so this also runs without the tree."""
    have = set(open(os.path.join(ROOT, "tests", "golden", "reference_utils_names.txt")).read().split())
    missing = [n for n in _utils_names_used_by_the_mirror() if n not in have]
    assert not missing, f"mirror uses utils names the reference's utils/general_utils.py does not define: {missing}"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussian_renderer")), reason="reference tree not present")
def test_b2_graft_mirror_runs_on_the_reference_utils(tmp_path):
    """the three mirror files dropped over the reference's (a symlinked package that shadows it) with the REFERENCE's
    utils/: import, partition a batch, build the band mask, stage the ground-truth bands (world size 1, CPU tensors)"""
    os.symlink(MIRROR, str(tmp_path / "gaussian_renderer"))
    code = (
        "import sys, types, torch\n"
        "import utils.general_utils as u\n"
        "assert u.__file__.startswith(%r), u.__file__\n"
        "assert not hasattr(u, 'device'), 'the reference utils has no device() helper'\n"
        "import gaussian_renderer as g, gaussian_renderer.workload_division as w, gaussian_renderer.loss_distribution as l\n"
        "assert g.__file__.startswith(%r), g.__file__\n"
        "a = types.SimpleNamespace(bsz=2, local_sampling=False, border_divpos_coeff=1.0, distributed_dataset_storage=False,\n"
        "    adjust_strategy_warmp_iterations=-1, no_heuristics_update=False, heuristic_decay=0.0, save_strategy_history=False,\n"
        "    gaussians_distribution=True, image_distribution=True, log_interval=250, log_folder='/tmp/x', zhx_debug=False,\n"
        "    zhx_time=False, lambda_dssim=0.2, lr_scale_loss=1.0)\n"
        "u.set_args(a); u.set_block_size(16, 16, 256); u.set_img_size(144, 208); u.set_cur_iter(1); u.init_distributed(a)\n"
        "u.GLOBAL_RANK = 0\n"
        "cams = [types.SimpleNamespace(uid=k, image_height=144, image_width=208,\n"
        "        original_image_backup=torch.full((3, 144, 208), k, dtype=torch.uint8), original_image=None) for k in range(2)]\n"
        "hist = w.DivisionStrategyHistoryFinal(types.SimpleNamespace(cameras=cams), 1, 0)\n"
        "st, tasks = w.start_strategy_final(cams, hist)\n"
        "assert tasks == [[(0, 0, 9), (1, 0, 9)]], tasks\n"
        "m = st[0].get_compute_locally(); assert m.shape == (9, 13) and bool(m.all())\n"
        "l.load_camera_from_cpu_to_all_gpu(cams, st, tasks)\n"
        "assert cams[1].original_image.shape == (3, 144, 208) and int(cams[1].original_image[0, 0, 0]) == 1\n"
        "l.load_camera_from_cpu_to_all_gpu_for_eval(cams, st, tasks)\n"
        "ca = g.get_cuda_args_final(st[0], 'train'); assert ca['mp_world_size'] == '1' and ca['stats_collector'] == {}\n"
        "stats = [{'forward_render_time': 1.0, 'backward_render_time': 2.0, 'forward_loss_time': 0.5} for _ in cams]\n"
        "w.finish_strategy_final(cams, hist, st, stats)\n"
        "assert hist.history[0]['all_gpu_running_time'] == [8.0]\n"
        "print('ok')\n" % (REF, str(tmp_path)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), REF, GRAFT]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
