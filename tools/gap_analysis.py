"""Where does the GPU idle inside a training iteration?  From a rocprofv3 kernel trace (rocpd sqlite) of bench.py:
take the steady-state iterations (delimited by the optimizer kernel: adam_multi_kernel, or the fused
preprocess_backward_adam kernel when K11 runs inside the step), and for every kernel print its mean duration and
the mean idle gap on the device BEFORE it (start - previous kernel's end, all queues merged).
Usage: python tools/gap_analysis.py results.db [skip_iterations]"""
import sqlite3
import sys
from collections import OrderedDict


WIDE = "--wide" in sys.argv


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:200] if WIDE else n.split("(")[0][:44]


def main():
    db = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    rows = c.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()
    # iterations: from the end of one optimizer kernel to the end of the next
    its, cur = [], []
    for name, s, e in rows:
        cur.append((short(name), s, e))
        if "adam_multi_kernel" in name or "preprocess_backward_adam" in name:
            its.append(cur)
            cur = []
    its = its[skip:]
    if not its:
        print("no iterations found")
        return
    lens = {}
    for it in its:
        lens[len(it)] = lens.get(len(it), 0) + 1
    common = max(lens, key=lens.get)
    its = [it for it in its if len(it) == common]
    print(f"{len(its)} iterations of {common} kernels each")
    n = len(its)
    tot_k = tot_g = 0.0
    print(f"{'#':>3s} {'kernel':44s} {'dur_us':>8s} {'gap_before_us':>13s}")
    for i in range(common):
        dur = sum(it[i][2] - it[i][1] for it in its) / n / 1e3
        if i == 0:
            gap = 0.0
        else:
            gap = sum(max(0, it[i][1] - max(x[2] for x in it[:i])) for it in its) / n / 1e3
        tot_k += dur
        tot_g += gap
        print(f"{i:3d} {its[0][i][0]:44s} {dur:8.2f} {gap:13.2f}" if not WIDE else f"{i:3d} {dur:8.2f} {gap:8.2f}  {its[0][i][0]}")
    span = sum(it[-1][2] - it[0][1] for it in its) / n / 1e3
    print(f"sum of kernel durations {tot_k:.1f} us, sum of idle gaps {tot_g:.1f} us, first-start to last-end {span:.1f} us")


if __name__ == "__main__":
    main()
