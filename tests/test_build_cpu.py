"""Register budgets of the hot kernels, checked at build time (no GPU needed: hipcc cross-compiles gfx950).

A kernel that silently starts to spill loses its occupancy class: in round 5 eight lines added to K10's worker loop
pushed it past its 128 registers (four workgroups per CU) -- 196 bytes of scratch, +20 % on every launch -- and nothing
but a benchmark on the GPU box showed it.  This test compiles the two translation units to assembly and reads the
kernels' resource notes."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "grendel-gs_amd", "csrc")
sys.path.insert(0, CSRC)

# kernel name fragment -> most VGPRs it may use (its occupancy class); every listed kernel must be free of spills
BUDGET = {
    "composite.hip": {"composite_backward_kernel": 128, "composite_forward_kernelILb0ELb0": 64,
                      "composite_forward_kernelILb1ELb1": 96},
    "binning.hip": {"emit_scatter_kernel": 64, "radix_onesweep_kernelILi8ELi512": 64, "bin_sort_persist_kernel": 128,
                    "bin_prepare_persist_kernelILi4": 128, "touch_count_kernel": 128},
}


def _resources(unit, tmp_path):
    import build as B  # csrc/build.py: the flags the library is built with

    out = os.path.join(str(tmp_path), unit + ".s")
    cmd = [B._hipcc()] + [f for f in B.FLAGS if f != "-fPIC"] + ["--cuda-device-only", "-S", os.path.join(CSRC, unit), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail(f"hipcc -S failed on {unit}:\n{r.stderr[-2000:]}")
    kernels, cur = {}, {}
    for line in open(out):
        m = re.match(r"\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        cur[m.group(1)] = m.group(2)
        if len(cur) == 5:
            kernels[cur["name"]] = {k: int(v) for k, v in cur.items() if k != "name"}
            cur = {}
    return kernels


@pytest.mark.parametrize("unit", sorted(BUDGET))
def test_hot_kernels_keep_their_register_budget(unit, tmp_path):
    kernels = _resources(unit, tmp_path)
    assert kernels, "no kernel resource notes found in the assembly"
    for frag, budget in BUDGET[unit].items():
        hits = {n: r for n, r in kernels.items() if frag in n}
        assert hits, f"{unit}: no kernel matches {frag!r} (renamed? update tests/test_build_cpu.py)"
        for name, r in hits.items():
            assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, \
                f"{name} spills {r['vgpr_spill_count']} registers ({r['private_segment_fixed_size']} B of scratch per lane)"
            assert r["vgpr_count"] <= budget, f"{name} uses {r['vgpr_count']} VGPRs, its occupancy class allows {budget}"
