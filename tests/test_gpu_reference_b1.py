"""Graft level B1 at RUN time: the reference's own Python on this repo's HIP operator, against the B2 mirror.

* `test_mirror_matches_frozen_reference_run`: tests/golden/reference_b1_*.npz hold what the REFERENCE's
  start_strategy_final / load_camera_from_cpu_to_all_gpu / distributed_preprocess3dgs_and_all2all_final / render_final /
  batched_loss_computation / finish_strategy_final / GaussianModel.training_setup + optimizer step returned on this
  operator (tests/golden/make_reference_b1_golden.py, run on an MI355X with the reference tree staged).  The mirror
  (+ fused K1 activations, fused exchange, fused loss, fused Adam) must reproduce them: cut points, task lists and
  exchange sizes exactly, received rows in the same order, image / loss / all six parameter gradients /
  means2D.grad to 1e-5, parameters after the optimizer step.  Runs on any GPU box.
* `test_reference_python_live`: the same comparison with the reference executed on the spot (more cases: world
  sizes 2 and 4 sharing the device over gloo, 1080p, the load balancer's heuristic trajectory under deterministic
  stand-in timings).  Needs the staged tree (tools/stage_reference.py) -> skipped on the driver's box.
* `test_reference_train_py_runs_unchanged`: `train.py` itself, unmodified, on a generated Blender-format scene.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "refgraft"))
import harness  # noqa: E402
import scenes  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
# norm-wise relative tolerance on image / loss / gradients of a ONE-iteration case: 1e-5 (3e-5 for the quaternion
# gradient only, see harness.compare)
TOL = {}
LOG = os.path.join(ROOT, "gpurun_out", "reference_b1_report.txt")


def _report(title, lines):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(f"== {title}\n" + "\n".join(lines) + "\n")


def _load_fixture(name):
    path = os.path.join(GOLDEN, f"reference_b1_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated yet")
    z = dict(np.load(path))
    world = int(z["world"])
    scene = {k[len("scene__"):]: v for k, v in z.items() if k.startswith("scene__")}
    outs = [{k[len(f"r{r}__"):]: v for k, v in z.items() if k.startswith(f"r{r}__")} for r in range(world)]
    for o in outs[1:]:
        o.setdefault("images", outs[0].get("images"))
    return z, scene, outs


@pytest.mark.parametrize("name", ["c0", "w2"])
def test_mirror_matches_frozen_reference_run(device, name):
    _, scene, ref = _load_fixture(name)
    mir = harness.run_side("mirror", scene)
    _report(f"mirror vs frozen reference run [{name}]", harness.compare(ref, mir, tol=TOL.get(name, 1e-5)))


@pytest.mark.parametrize("name", ["hd", "hdw2", "w2b2", "w4b2", "w8b4"])
def test_mirror_matches_frozen_reference_summary(device, name):
    z, _, summ = _load_fixture(name)
    scene = scenes.build_case(name)
    chk = sum(float(np.asarray(v, np.float64).sum()) for v in scene.values())
    assert abs(chk - float(z["scene_checksum"])) <= 1e-9 * abs(chk), "the seeded scene is not the frozen one"
    mir = harness.run_side("mirror", scene)
    tol = TOL.get(name, 1e-5) if int(scene["iters"]) == 1 else 2e-3
    _report(f"mirror vs frozen reference summary [{name}]", harness.compare_summary(summ, mir, tol=tol))


@pytest.mark.skipif(not harness.reference_staged(), reason="reference tree not staged (tools/stage_reference.py)")
@pytest.mark.parametrize("name", ["c0", "w2", "w2b2", "w4b2", "w8b4", "hd", "hdw2"])
def test_reference_python_live(device, name):
    scene = scenes.build_case(name)
    ref = harness.run_side("ref", scene)
    mir = harness.run_side("mirror", scene)
    _report(f"reference python live vs mirror [{name}]", harness.compare(ref, mir, tol=TOL.get(name, 1e-5)))


@pytest.mark.skipif(not harness.reference_staged(), reason="reference tree not staged (tools/stage_reference.py)")
def test_reference_train_py_runs_unchanged(device, tmp_path):
    """the reference's train.py, byte for byte, with grendel-gs_amd/b1_graft on the module path: dataset loading,
    GaussianModel.create_from_pcd (simple_knn shim), 60 iterations incl. two densification rounds, evaluation and
    the final save.  The loss must fall and the evaluation PSNR must be finite."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_blender_scene

    data = str(tmp_path / "matrixcity_synth")
    make_blender_scene.generate(data, n_views=12, width=208, height=144)
    model = str(tmp_path / "model")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([harness.REF_ROOT, os.path.join(ROOT, "grendel-gs_amd", "b1_graft")]))
    cmd = [sys.executable, os.path.join(harness.REF_ROOT, "train.py"), "-s", data, "--model_path", model, "--preload_dataset_to_gpu",
           "--iterations", "60", "--densify_from_iter", "20", "--densification_interval", "20",
           "--test_iterations", "60", "--save_iterations", "60", "--log_interval", "10", "--eval"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=harness.REF_ROOT)
    tail = (r.stdout + r.stderr)[-6000:]
    assert r.returncode == 0, tail
    log = open(os.path.join(model, "python_ws=1_rk=0.log")).read()
    _report("reference train.py unchanged (60 iterations, 208x144, W=1)", [tail, "---- python_ws=1_rk=0.log (tail)",
                                                                          log[-3000:]])
    assert "Training complete" in r.stdout
    import re

    epochs = [float(x) for x in re.findall(r"epoch \d+ loss: ([0-9.eE+-]+)", log)]
    assert len(epochs) >= 4 and epochs[-1] < 0.95 * epochs[0], f"the loss does not fall: {epochs}"
    psnr = [float(x) for x in re.findall(r"Evaluating test: L1 [0-9.eE+-]+ PSNR ([0-9.eE+-]+)", r.stdout + log)]
    assert psnr and all(5.0 < p < 60.0 for p in psnr), psnr
    assert "Number of split gaussians" in log, "densification ran on means2D.grad of the HIP operator"
    assert os.path.isdir(os.path.join(model, "point_cloud")), "the final save wrote the point cloud"


@pytest.mark.skipif(not harness.reference_staged(), reason="reference tree not staged (tools/stage_reference.py)")
def test_cut_points_equal_the_reference_function(device):
    """a13: the mirror computes the prefix sums of the per-row costs on the HOST (no device sync per iteration); the
    reference's division_pos_heuristic does it on the device.  400 seeded cost vectors (time-like plateaus, smooth,
    spiky, all-ones; 2 / 4 / 8 ranks; 1080p, 4K and batched row counts): identical cut points."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([harness.REF_ROOT, os.path.join(ROOT, "grendel-gs_amd", "b1_graft")]))
    n = 400
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "refgraft", "cuts_fuzz.py"), str(n)], env=env,
                       capture_output=True, text=True, timeout=600, cwd=harness.REF_ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    ref = json.loads(r.stdout.strip().splitlines()[-1])
    import torch

    from gaussian_renderer.workload_division import division_pos_heuristic

    bad = []
    for seed in range(n):
        rows, world = scenes.fuzz_case(seed)
        mine = division_pos_heuristic(torch.from_numpy(scenes.fuzz_heuristics(seed, rows)), rows, world, right=True)
        if mine != ref[seed]:
            bad.append((seed, rows, world, mine, ref[seed]))
    _report("cut points vs the reference's division_pos_heuristic", [f"{n} cases, {len(bad)} differ"] + [str(b) for b in bad[:5]])
    assert not bad, bad[:5]
