"""average shader clock per kernel from a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE run: cycles / duration.
usage: python3 tools/pmc_clock.py <counter_collection.csv> <kernel_trace.csv> [name substring ...]"""
import csv
import sys
from collections import defaultdict

cc, kt = sys.argv[1], sys.argv[2]
subs = sys.argv[3:] or [""]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    d = dur.get(r["Dispatch_Id"])
    if not d:
        continue
    name = d[1]
    if not any(s in name for s in subs):
        continue
    a = acc[name[:60]]
    a[0] += float(r["Counter_Value"])
    a[1] += d[0]
    a[2] += 1
for k, (c, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} n={n:5d} avg {ns / n / 1e3:8.1f} us  cycles/launch {c / n:12.0f}  clock {c / ns:6.3f} GHz (if the counter sums XCDs: {c / ns / 8:6.3f})")
