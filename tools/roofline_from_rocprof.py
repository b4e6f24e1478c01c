"""Recomputes the rocprofv3-side dense-family fraction from the committed summary (cross-check only: bench.py's roofline comes from the in-kernel timeline, tools/timeline.py):

    python tools/roofline_from_rocprof.py profiles/r02_kernel_stats.csv [launches_per_step] [algorithmic_gflop_per_step]

frac_rocprof = (algorithmic FLOPs of one step / dense launches of one step) / (launch-weighted average rocprofv3 duration of the
forward dense kernels) / 157.3 TFLOP/s.  Defaults: 746 launches and 43.34 GFLOP (cfg-2: 135.4 MFLOP/frame x 320 frames)."""
import csv
import sys

PEAK = 157.3e12


def dense_average_ns(path):
    tot = calls = 0.0
    per = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("void ", "")
        if name.startswith(("k_linear<", "k_linear_what<", "k_linear_rows", "k_linear_mt")):
            tot += float(r["TotalDurationNs"])
            calls += float(r["Calls"])
            per[name.split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]))
    return tot / calls, per


if __name__ == "__main__":
    avg_ns, per = dense_average_ns(sys.argv[1])
    launches = float(sys.argv[2]) if len(sys.argv) > 2 else 746.0
    gflop = float(sys.argv[3]) if len(sys.argv) > 3 else 2 * 5 * 13543744 * 320 / 1e9
    achieved = gflop * 1e9 / launches / (avg_ns * 1e-9)
    dom = max(per, key=lambda k: per[k][0])
    print("dense family: average launch %.3f us (dominant %s: %.3f us over %d launches)" % (avg_ns / 1e3, dom, per[dom][1] / 1e3, per[dom][0]))
    print("algorithmic %.2f MFLOP per launch -> %.2f TFLOP/s -> frac_rocprof %.4f" % (gflop * 1e3 / launches, achieved / 1e12, achieved / PEAK))
