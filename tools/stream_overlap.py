"""Do the views of bench.py's two-stream render legs overlap on the device?  From a rocprofv3 kernel trace (rocpd database) of
    rocprofv3 --kernel-trace -d DIR -- python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 2 --repeats 1 --render-steps N
the last N views are the pipelined two-stream leg, the N before them the plain two-stream leg, the N before those the
sequential one-stream leg (warm-up views of a leg are attributed to it).  For each leg: wall span of its kernels, the sum of
their durations, the time during which kernels of TWO queues ran at once, and the pairs that overlapped most.
Usage: python tools/stream_overlap.py DIR_OR_DB N > profiles/r06_two_stream_overlap.txt"""
import glob
import os
import sqlite3
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:28]


def main():
    p, n_views = sys.argv[1], int(sys.argv[2])
    db = p if p.endswith(".db") else sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True), key=os.path.getsize)[-1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    last_bwd = max((e for n, s, e, q in rows if "composite_backward" in n), default=0)
    tail = [(n, s, e, q) for n, s, e, q in rows if s > last_bwd]  # the forward-only legs
    k8 = [i for i, r in enumerate(tail) if "composite_forward" in r[0]]
    print(f"{len(rows)} kernel dispatches, {len(tail)} after the last training step, {len(k8)} views rendered there")
    legs = [("two streams, pipelined", k8[-n_views], k8[-1] + 1)]  # (up to the last view's composite kernel)
    if len(k8) >= 2 * n_views + 4:
        legs.insert(0, ("two streams", k8[-2 * n_views - 4], k8[-n_views - 4]))
    if len(k8) >= 3 * n_views + 8:
        legs.insert(0, ("one stream", k8[-3 * n_views - 8], k8[-2 * n_views - 8]))
    for title, a, b in legs:
        ks = tail[a:b]
        if not ks:
            continue
        t0, t1 = min(s for _, s, _, _ in ks), max(e for _, _, e, _ in ks)
        busy = sum(e - s for _, s, e, _ in ks)
        queues = sorted({q for _, _, _, q in ks})
        ev = sorted([(s, 1, i) for i, (_, s, _, _) in enumerate(ks)] + [(e, -1, i) for i, (_, _, e, _) in enumerate(ks)])
        active, prev, two, any_ = set(), t0, 0, 0
        pair = {}
        for t, d, i in ev:
            if active:
                any_ += t - prev
                qs = {ks[j][3] for j in active}
                if len(qs) >= 2:
                    two += t - prev
                    names = sorted({short(ks[j][0]) for j in active})
                    key = " || ".join(names[:3])
                    pair[key] = pair.get(key, 0) + (t - prev)
            prev = t
            (active.add if d > 0 else active.discard)(i)
        views = sum(1 for n, _, _, _ in ks if "composite_forward" in n)
        print(f"\n{title}: {views} views on queues {queues}; span {(t1 - t0) / 1e6:.3f} ms = {(t1 - t0) / 1e3 / max(views, 1):.1f} us per view; "
              f"kernel time {busy / 1e6:.3f} ms; device busy {any_ / 1e6:.3f} ms; two queues at once {two / 1e6:.3f} ms "
              f"({100.0 * two / max(t1 - t0, 1):.1f} % of the span)")
        for k, v in sorted(pair.items(), key=lambda kv: -kv[1])[:8]:
            print(f"    {v / 1e3:9.1f} us  {k}")


if __name__ == "__main__":
    main()
