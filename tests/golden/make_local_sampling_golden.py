"""Freeze what the REFERENCE's start_strategy_final returns in `--local_sampling` mode
(gaussian_renderer/workload_division.py:858-877) -> tests/golden/reference_local_sampling.json.

Runs in the build container (host-only code path): imports the reference's gaussian_renderer.workload_division from
/root/reference with grendel-gs_amd/b1_graft on the module path (the import shim of the rasterizer package), sets the
process globals the function reads (utils/general_utils.py: ARGS, WORLD_SIZE, GLOBAL_RANK, DEFAULT_GROUP, TILE_Y) and
records, for a few (world, bsz, rank) settings, the task lists and every strategy's fields.  The mirror is held to it by
tests/test_dist_cpu.py::test_local_sampling_strategy_equals_the_reference."""
import json
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
CASES = [(2, 2, 0), (2, 4, 1), (4, 8, 2), (8, 8, 5), (8, 16, 7)]  # (world, bsz, rank)


def main():
    sys.path[:0] = [REF, os.path.join(ROOT, "grendel-gs_amd", "b1_graft")]
    import gaussian_renderer.workload_division as wd
    import utils.general_utils as utils

    assert wd.__file__.startswith(REF) and utils.__file__.startswith(REF)
    out = []
    for world, bsz, rank in CASES:
        utils.ARGS = SimpleNamespace(local_sampling=True, bsz=bsz)
        utils.WORLD_SIZE, utils.GLOBAL_RANK = world, rank
        utils.DEFAULT_GROUP = SimpleNamespace(size=lambda w=world: w, rank=lambda r=rank: r)
        utils.IMG_H, utils.IMG_W, utils.TILE_Y, utils.TILE_X = 1080, 1920, 68, 120
        cams = [SimpleNamespace(uid=100 + k) for k in range(bsz)]
        strategies, tasks = wd.start_strategy_final(cams, None)
        out.append({"world": world, "bsz": bsz, "rank": rank, "tile_y": 68,
                    "tasks": [[list(t) for t in g] for g in tasks],
                    "strategies": [{"uid": s.camera.uid, "world_size": s.world_size, "gpu_ids": list(s.gpu_ids),
                                    "division_pos": list(s.division_pos), "rank": s.rank} for s in strategies]})
    path = os.path.join(ROOT, "tests", "golden", "reference_local_sampling.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
