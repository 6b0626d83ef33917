"""Per-kernel average launch durations from a `rocprofv3 --kernel-trace --stats` run of bench.py --serial-lanes:

    python tools/kstats_summary.py <kernel_stats.csv> <out.json>

Writes {source_sha, kernels: {name: {calls, avg_us, min_us, max_us}}} with bench.py's GEMM variant names (the ones its HIP-event
roofline leg reports), so that bench.py can print the profiler's clock for its dominant kernel beside its own (`rocprof_avg_us`,
`frac_rocprof`) whenever the summary was collected on the kernel sources the shipped library was built from."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_traffic_summary import short          # noqa: E402
import bench                                          # noqa: E402


def main():
    src, dst = sys.argv[1], sys.argv[2]
    acc = {}
    with open(src, newline="") as f:
        for r in csv.DictReader(f):
            n = short(r["Name"])
            calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
            a = acc.setdefault(n, {"calls": 0, "total_ns": 0.0, "min_us": 1e30, "max_us": 0.0})
            a["calls"] += calls
            a["total_ns"] += tot
            a["min_us"] = min(a["min_us"], float(r["MinNs"]) / 1e3)
            a["max_us"] = max(a["max_us"], float(r["MaxNs"]) / 1e3)
    out = {"source_sha": bench.source_sha(), "from": os.path.basename(src),
           "kernels": {n: {"calls": a["calls"], "avg_us": round(a["total_ns"] / a["calls"] / 1e3, 3), "min_us": round(a["min_us"], 3),
                           "max_us": round(a["max_us"], 3)} for n, a in acc.items()}}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    top = sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"])[:8]
    for n, a in top:
        print("%-44s calls %5d  avg %8.2f us" % (n[:44], a["calls"], a["avg_us"]))


if __name__ == "__main__":
    main()
