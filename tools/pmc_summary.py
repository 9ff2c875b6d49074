"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md prescribes: TCC slot limits), calibrated on a device copy of known
size contained in the same runs (tools/kbench.py --calib: 256 MiB read + 256 MiB written per copy).
Usage: python tools/pmc_summary.py fetch.db write.db > profiles/rNN_pmc_traffic.txt  (also writes
profiles/pmc_traffic.json for bench.py's roofline.traffic)."""
import json
import os
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                     "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    MiB256 = 256 * 1024 * 1024
    calib_name = None
    for k in fetch:  # the calibration multiply: ~256 MiB written per call
        if "vectorized_elementwise_kernel" in k and k in write and abs(write[k][1] * 1024 / MiB256 - 1.0) < 0.05:
            calib_name = k
    f_scale = w_scale = None
    if calib_name:
        f_scale = MiB256 / (fetch[calib_name][1] * 1024)
        w_scale = MiB256 / (write[calib_name][1] * 1024)
        print(f"calibration kernel: {calib_name[:80]}")
        print(f"  FETCH_SIZE reports {fetch[calib_name][1] / 1024:.1f} MiB for a 256 MiB streaming read  -> x{f_scale:.3f}")
        print(f"  WRITE_SIZE reports {write[calib_name][1] / 1024:.1f} MiB for a 256 MiB streaming write -> x{w_scale:.3f}")
        print("  (gather-dominated kernels -- composite_* -- read 8-16 B per lane at random addresses: their true read")
        print("   traffic lies between the raw FETCH_SIZE and the scaled value; hbm_MB(cal) is the scaled upper bound)")
    out = {}
    print(f"{'kernel':70s} {'calls':>6s} {'fetch_MB':>10s} {'write_MB':>10s} {'hbm_MB(cal)':>12s}")
    names = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1]))
    for k in names[:40]:
        f = fetch.get(k, (0, 0.0))
        w = write.get(k, (0, 0.0))
        fb = f[1] * 1024 * (f_scale or 1.0)
        wb = w[1] * 1024 * (w_scale or 1.0)
        short = k.replace("(anonymous namespace)::", "").split("(")[0][:70]
        print(f"{short:70s} {f[0]:6d} {f[1] * 1024 / 1e6:10.2f} {w[1] * 1024 / 1e6:10.2f} {(fb + wb) / 1e6:12.2f}")
        out[short] = {"hbm_bytes_per_launch": int(fb + wb), "fetch_raw_bytes": int(f[1] * 1024),
                      "write_raw_bytes": int(w[1] * 1024), "fetch_scale": f_scale, "write_scale": w_scale}
    alias = {"composite_backward": "composite_backward_kernel", "composite_forward": "composite_forward_kernel",
             "preprocess_forward": "void preprocess_forward_kernel<3>", "preprocess_backward": "void preprocess_backward_kernel<3>"}
    js = {}
    for short, full in alias.items():
        if full in out:
            js[short] = out[full]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    js["_note"] = ("per launch, 1 M Gaussians / 1920x1080 / view 0 (tools/kbench.py); FETCH_SIZE and WRITE_SIZE from "
                   "separate rocprofv3 --pmc passes, scaled by the factors measured on a 256 MiB device copy in the same run")
    json.dump(js, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
