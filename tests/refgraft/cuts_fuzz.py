"""Reference side of the cut-point fuzz: the REFERENCE's division_pos_heuristic (cumsum + searchsorted on the device,
gaussian_renderer/workload_division.py:75-94) on seeded per-row costs.  Run with the staged reference tree + b1_graft on
PYTHONPATH; prints one JSON list.  Test infrastructure."""
import json
import sys

import numpy as np
import torch

import gaussian_renderer.workload_division as wd


def heuristics(seed, rows):
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


def main():
    n_cases = int(sys.argv[1])
    out = []
    for seed in range(n_cases):
        rows = [68, 135, 68 * 4, 35 * 4][seed % 4]
        world = [2, 4, 8, 8][(seed // 4) % 4]
        h = torch.from_numpy(heuristics(seed, rows)).cuda()
        out.append(wd.division_pos_heuristic(h, rows, world, right=True))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
