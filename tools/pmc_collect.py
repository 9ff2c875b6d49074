"""Turn rocprofv3 runs of `bench.py` into profiles/rNN_pmc.json + a readable per-kernel table.

Inputs: rocpd sqlite databases written by (each in its OWN run, as MI355X_MICROARCH.md prescribes: TCC slot limits,
and never combined with the hip / hsa / memory-copy trace domains):
    rocprofv3 --kernel-trace                      -d DIR_TRACE -- python bench.py --no-cpu-baseline ...
    rocprofv3 --kernel-trace --pmc SQ_...(<= 8)   -d DIR_SQ    -- python bench.py --no-cpu-baseline ...
    rocprofv3 --kernel-trace --pmc FETCH_SIZE     -d DIR_FETCH -- python bench.py --no-cpu-baseline --pmc-calib ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE     -d DIR_WRITE -- python bench.py --no-cpu-baseline --pmc-calib ...
Usage: python tools/pmc_collect.py --trace DIR --sq DIR --fetch DIR --write DIR --out profiles/r02_pmc.json \
                                   --command "<the bench.py command line>" > profiles/r02_pmc.txt

Kernel symbols are grouped the way bench.py's `kernels` block groups them (one group = one C-ABI call); a group's
per-launch figure = sum over its member kernels / number of calls.  HBM bytes: FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half of a wide coalesced stream, so both are scaled by the factors measured on the
256 MiB streaming multiply contained in the same runs (--pmc-calib); for gather-dominated kernels the scaled read
figure is an upper bound.  VALU: SQ_ACTIVE_INST_VALU is in quad-cycles summed over the SIMDs, SQ_BUSY_CYCLES in
cycles summed over the 32 shader engines: valu_busy = 4 * ACTIVE / (BUSY / 32 * 1024 SIMDs)."""
import argparse
import glob
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROUPS = [  # (group, substrings of the kernel symbol, substring that marks ONE call of the group)
    ("composite_forward", ["composite_forward_kernel"], "composite_forward_kernel"),
    ("composite_backward", ["composite_backward_kernel"], "composite_backward_kernel"),
    ("preprocess_forward", ["preprocess_forward"], "preprocess_forward"),
    ("preprocess_backward_adam", ["preprocess_backward_adam"], "preprocess_backward_adam"),  # fused K11 + Adam (first)
    ("preprocess_backward", ["preprocess_backward"], "preprocess_backward"),
    # (one call = one K3: the look-back pipeline's touch_count_kernel or the persistent prepare kernel of round 5)
    ("binning", ["touch_count_kernel", "radix_onesweep_kernel", "scan_gather_lookback_kernel", "emit_scatter_kernel",
                 "emit_pairs_kernel", "tile_ranges", "bin_prepare_persist_kernel", "bin_sort_persist_kernel",
                 "seg_scatter_kernel", "seg_scan_kernel", "tile_base_kernel", "pair_scatter_kernel"],
     ("touch_count_kernel", "bin_prepare_persist_kernel")),
    ("l1_ssim_forward", ["l1_ssim_forward_kernel", "l1_ssim_finalize_kernel"], "l1_ssim_forward_kernel"),
    ("l1_ssim_backward", ["l1_ssim_backward_kernel"], "l1_ssim_backward_kernel"),
    ("adam", ["adam_kernel", "adam_multi_kernel"], None),
]
MiB256 = 256 * 1024 * 1024


def find_db(d):
    if not d:
        return None
    if d.endswith(".db"):
        return d
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True), key=os.path.getsize)
    return dbs[-1] if dbs else None


def kernel_times(db):
    c = sqlite3.connect(db)
    return {r[0]: (r[1], r[2], r[3], r[4], r[5]) for r in c.execute(
        "select name, count(*), sum(duration), max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name")}


def counter_sums(db):
    """kernel symbol -> counter -> (dispatches, sum of values, sum of durations ns)"""
    c = sqlite3.connect(db)
    out = {}
    for name, ctr, n, v, dur in c.execute(
            "select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
            "group by kernel_name, counter_name"):
        out.setdefault(name, {})[ctr] = (n, v, dur)
    return out


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "grendel-gs_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def group_of(sym):
    for g, subs, _ in GROUPS:
        if any(s in sym for s in subs):
            return g
    return None


def main():
    ap = argparse.ArgumentParser()
    for k in ("trace", "sq", "fetch", "write"):
        ap.add_argument("--" + k, default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_pmc.json"))
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    res = {"source_hash": source_hash(), "command": a.command, "kernels": {}, "calibration": {}}
    K = res["kernels"]

    def calls_of(table, g):
        marker = [m for gg, _, m in GROUPS if gg == g][0]
        if marker is None:
            return sum(n for sym, (n, *_) in table.items() if group_of(sym) == g)
        markers = (marker,) if isinstance(marker, str) else marker
        return sum(n for sym, (n, *_) in table.items() if any(m in sym for m in markers))

    tdb = find_db(a.trace)
    if tdb:
        kt = kernel_times(tdb)
        print(f"# kernel trace: {tdb}")
        print(f"{'kernel symbol':78s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s}")
        for sym, (n, dur, vg, sg, lds) in sorted(kt.items(), key=lambda kv: -kv[1][1])[:32]:
            short = sym.replace("(anonymous namespace)::", "").split("(")[0][:78]
            print(f"{short:78s} {n:6d} {dur / 1e6:9.3f} {dur / n / 1e3:9.2f} {vg or 0:5d} {sg or 0:5d} {lds or 0:6d}")
        for g, _, _ in GROUPS:
            members = {s: v for s, v in kt.items() if group_of(s) == g}
            if not members:
                continue
            calls = calls_of(kt, g)
            K.setdefault(g, {})["trace_calls"] = calls
            K[g]["trace_avg_ms"] = round(sum(v[1] for v in members.values()) / calls / 1e6, 5)
            K[g]["trace_kernels_per_call"] = round(sum(v[0] for v in members.values()) / calls, 2)
        print()

    sdb = find_db(a.sq)
    if sdb:
        cs = counter_sums(sdb)
        print(f"# SQ counters: {sdb}")
        for g, _, _ in GROUPS:
            members = {s: v for s, v in cs.items() if group_of(s) == g}
            if not members:
                continue
            anyc = next(iter(next(iter(members.values())).keys()))
            table = {s: (v[anyc][0],) for s, v in members.items()}
            calls = calls_of(table, g)
            d = K.setdefault(g, {})
            ctrs = sorted({c for v in members.values() for c in v})
            for c in ctrs:
                d[c] = round(sum(v[c][1] for v in members.values() if c in v) / calls, 1)
            d["avg_ms"] = round(sum(v[anyc][2] for v in members.values()) / calls / 1e6, 5)  # duration in the PMC run
            if d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_BUSY_CYCLES"):
                d["valu_busy_frac"] = round(4.0 * d["SQ_ACTIVE_INST_VALU"] / (d["SQ_BUSY_CYCLES"] / 32.0 * 1024.0), 4)
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and d.get("SQ_BUSY_CYCLES"):
                # cycles the matrix pipe is busy, summed over the SIMDs (fp32 MFMA does not co-issue with the VALU on
                # this chip: SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r02_composite_sq_counters.txt)
                d["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["SQ_BUSY_CYCLES"] / 32.0 * 1024.0), 4)
            if d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_INSTS_VALU"):
                d["cycles_per_valu_inst"] = round(4.0 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"], 3)
            print(f"{g:22s} calls {calls:4d}  avg {d['avg_ms']:.4f} ms  " +
                  "  ".join(f"{c}={d[c]:.3g}" for c in ctrs) +
                  f"  valu_busy={d.get('valu_busy_frac')}  cyc/inst={d.get('cycles_per_valu_inst')}")
        print()

    scale = {}
    for key, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        db = find_db(getattr(a, key))
        if not db:
            continue
        cs = counter_sums(db)
        print(f"# {ctr}: {db}")
        # calibration: the 256 MiB streaming multiply (vectorized_elementwise_kernel launched 3 times by --pmc-calib)
        cal = None
        for sym, v in cs.items():
            if "vectorized_elementwise_kernel" in sym and ctr in v and v[ctr][0] == 3:
                per = v[ctr][1] / v[ctr][0] * 1024.0
                if 0.4 < per / MiB256 < 1.1:
                    cal = per
        scale[key] = (MiB256 / cal) if cal else (2.0 if key == "fetch" else 1.0)
        res["calibration"][key] = {"reported_bytes_for_256MiB": cal, "scale": scale[key],
                                   "assumed": cal is None}
        print(f"  calibration: {ctr} reports {cal} B for a 256 MiB stream -> x{scale[key]:.3f}"
              f"{' (ASSUMED, no calibration kernel found)' if cal is None else ''}")
        for g, _, _ in GROUPS:
            members = {s: v for s, v in cs.items() if group_of(s) == g and ctr in v}
            if not members:
                continue
            table = {s: (v[ctr][0],) for s, v in members.items()}
            calls = calls_of(table, g)
            raw = sum(v[ctr][1] for v in members.values()) / calls * 1024.0
            d = K.setdefault(g, {})
            d[f"{key}_raw_bytes"] = int(raw)
            d[f"{key}_bytes"] = int(raw * scale[key])
            d.setdefault("avg_ms", round(sum(v[ctr][2] for v in members.values()) / calls / 1e6, 5))
            print(f"  {g:22s} calls {calls:4d}  raw {raw / 1e6:9.2f} MB  scaled {raw * scale[key] / 1e6:9.2f} MB")
        print()
    for g, d in K.items():
        if "fetch_bytes" in d and "write_bytes" in d:
            d["hbm_bytes_per_launch"] = d["fetch_bytes"] + d["write_bytes"]
    json.dump(res, open(a.out, "w"), indent=1)
    print(f"# wrote {a.out} (source hash {res['source_hash']})", file=sys.stderr)


if __name__ == "__main__":
    main()
