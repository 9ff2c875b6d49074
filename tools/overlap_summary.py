"""Concurrency of the exchange kernels with the composite kernels, from a rocprofv3 kernel trace of
tools/overlap_trace.py.  Usage: python tools/overlap_summary.py DIR_OR_DB > profiles/r02_overlap.txt"""
import glob
import os
import sqlite3
import sys

EXCH = ("exchange_pack_kernel", "exchange_count_kernel", "scatter_add_rows_kernel", "gather_rows_kernel", "ncclDevKernel",
        "rccl", "AllToAll", "SendRecv")
COMP = ("composite_forward_kernel", "composite_backward_kernel", "touch_count_kernel", "radix_onesweep", "emit_scatter",
        "l1_ssim", "preprocess_backward")


def main():
    p = sys.argv[1]
    db = p if p.endswith(".db") else sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True), key=os.path.getsize)[-1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
    ex = [(s, e, n, q) for n, s, e, st, q in rows if any(k in n for k in EXCH)]
    co = [(s, e, n, q) for n, s, e, st, q in rows if any(k in n for k in COMP)]
    print(f"{len(rows)} kernel dispatches; {len(ex)} exchange-side, {len(co)} render / loss / K11-side")
    queues = {}
    for n, s, e, st, q in rows:
        queues.setdefault(q, set()).add(n.replace("(anonymous namespace)::", "").split("(")[0][:40])
    for q, names in queues.items():
        print(f"  queue {q}: {len(names)} kernel names, e.g. {sorted(names)[:6]}")
    # overlap of every exchange kernel with render-side kernels (sweep)
    tot_ex = sum(e - s for s, e, _, _ in ex)
    ov = 0
    by = {}
    j = 0
    co_sorted = sorted(co)
    for s, e, n, q in ex:
        while j < len(co_sorted) and co_sorted[j][1] <= s:
            j += 1
        k = j
        while k < len(co_sorted) and co_sorted[k][0] < e:
            o = min(e, co_sorted[k][1]) - max(s, co_sorted[k][0])
            if o > 0:
                ov += o
                key = (n.replace("(anonymous namespace)::", "").split("(")[0][:32],
                       co_sorted[k][2].replace("(anonymous namespace)::", "").split("(")[0][:32])
                by[key] = by.get(key, 0) + o
            k += 1
    print(f"exchange-side kernel time {tot_ex / 1e6:.3f} ms, of which {ov / 1e6:.3f} ms "
          f"({100.0 * ov / max(tot_ex, 1):.1f} %) ran concurrently with render-side kernels")
    for (a, b), o in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {a:34s} || {b:34s} {o / 1e3:10.1f} us")


if __name__ == "__main__":
    main()
