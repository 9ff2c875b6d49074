"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per-kernel calls / total / mean / %,
the same table `--stats` prints.  Usage: python tools/rocpd_summary.py results.db [top_n] > profiles/x.txt"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
          f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>6s}")
    for r in rows[:top]:
        print(f"{r[0][:90]:90s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} "
              f"{100 * r[2] / total:6.2f} {r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:6d}")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main()
