"""Reference side of the cut-point fuzz: the REFERENCE's division_pos_heuristic (cumsum + searchsorted on the device,
gaussian_renderer/workload_division.py:75-94) on seeded per-row costs.  Run with the staged reference tree + b1_graft on
PYTHONPATH; prints one JSON list.  Test infrastructure."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gaussian_renderer.workload_division as wd  # noqa: E402  (the reference's)
import scenes  # noqa: E402


def main():
    out = []
    for seed in range(int(sys.argv[1])):
        rows, world = scenes.fuzz_case(seed)
        h = torch.from_numpy(scenes.fuzz_heuristics(seed, rows)).cuda()
        out.append(wd.division_pos_heuristic(h, rows, world, right=True))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
