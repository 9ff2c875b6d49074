"""TIMING PROBE, wrong results by construction: what would the one-sweep radix passes cost if a workgroup's look-back only
ever read state words written on its OWN XCD?  Workgroup i (dispatched to XCD i % 8) takes tile (i % 8) * R + i / 8 with
R = ceil(tiles / 8), so consecutive tiles run on one XCD, and the look-back stops at the start of the XCD's range (the
first tile of a range publishes a zero prefix) -- the eight chains are independent, the digit runs of different ranges
overwrite each other (all writes stay inside the digit's own run: memory-safe).  Compare with the production library:

    python tools/diag/radix_xcd_chain_probe.py build          (build container) -> variants/libgsraster_xcdchain.so
    python tools/binbench.py --lib variants/libgsraster_xcdchain.so            (GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "grendel-gs_amd", "csrc")


def _sub(src, old, new):
    assert src.count(old) == 1, (src.count(old), old[:80])
    return src.replace(old, new)


def build():
    rx = open(os.path.join(CSRC, "radix.h")).read()
    rx = _sub(rx, "    if (threadIdx.x == 0) sm.bid = atomicAdd(ticket, 1u);",
              "    if (threadIdx.x == 0) sm.bid = (blockIdx.x & 7u) * ((gridDim.x + 7u) / 8u) + (blockIdx.x >> 3);")
    rx = _sub(rx, "        if (live) st_agent(&row[d], bid == 0 ? (tot | LB_PRE) : (tot + 1u));",
              "        const uint32_t R__ = (gridDim.x + 7u) / 8u, rs__ = (bid / R__) * R__;\n"
              "        if (live) st_agent(&row[d], bid == rs__ ? (tot | LB_PRE) : (tot + 1u));")
    rx = _sub(rx, "        if (live && bid > 0) {", "        if (live && bid > rs__) {")
    rx = _sub(rx, "                    v[k] = (j - k >= 0) ? ld_agent(", "                    v[k] = (j - k >= (long long)rs__) ? ld_agent(")
    vd = os.path.join(ROOT, "variants")
    os.makedirs(vd, exist_ok=True)
    open(os.path.join(vd, "radix.h"), "w").write(rx)
    path = os.path.join(vd, "binning_xcdchain.hip")
    open(path, "w").write(open(os.path.join(CSRC, "binning.hip")).read())
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "xcdchain", "--src",
                        f"binning={path}"], check=True)
    finally:
        os.remove(os.path.join(vd, "radix.h"))


if __name__ == "__main__":
    build()
