"""Freeze what the REFERENCE's own Python returns on this repo's operator (graft level B1) -> tests/golden/reference_b1_*.npz

Runs on a GPU box that carries the staged reference tree (python tools/stage_reference.py, then gpurun):
tests/refgraft/run_iteration.py --side ref imports the reference's gaussian_renderer / scene / arguments / utils from
_refstage/reference and drives start_strategy_final -> load_camera_from_cpu_to_all_gpu ->
distributed_preprocess3dgs_and_all2all_final -> render_final -> batched_loss_computation -> backward ->
finish_strategy_final -> optimizer step (train_internal.py:134-208, 316-329) on the HIP operator.  What those
functions returned is stored next to the inputs; tests/test_gpu_reference_b1.py then holds the B2 mirror (rows
a10-a17 of SURVEY.md section 8) to it on any box, with or without the reference tree.  The fixtures pin the MIRROR on the
reference's host code; the rasterizer arithmetic underneath both is this repo's ("parity unpinned").
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "refgraft"))
import harness  # noqa: E402
import scenes  # noqa: E402

FULL = ["c0", "w2"]          # inputs + every output
SUMMARY = ["hd", "hdw2", "w2b2", "w4b2", "w8b4"]  # inputs regenerated from the seed (checksum stored), outputs summarised


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    assert harness.reference_staged(), "stage the reference first: python tools/stage_reference.py"
    only = sys.argv[2:]
    for name in [n for n in FULL + SUMMARY if not only or n in only]:
        scene = scenes.build_case(name)
        outs = harness.run_side("ref", scene)
        blob = {}
        if name in FULL:
            for k, v in scene.items():
                blob["scene__" + k] = v
        else:
            blob["scene_checksum"] = np.float64(sum(float(np.asarray(v, np.float64).sum()) for v in scene.values()))
            outs = harness.summarize(outs)
        for r, o in enumerate(outs):
            for k, v in o.items():
                if name in FULL and r > 0 and k == "images":
                    continue  # the SUM-assembled stack is the same on every rank
                blob[f"r{r}__{k}"] = v
        blob["world"] = np.int64(len(outs))
        path = os.path.join(out_dir, f"reference_b1_{name}.npz")
        np.savez_compressed(path, **blob)
        print(name, "->", path, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main()
