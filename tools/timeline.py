"""Timeline of ONE step from a rocprofv3 --kernel-trace csv (concurrent lanes): per kernel start/end relative to the step's first
launch, queue, and the idle gaps.  usage: python tools/timeline.py <kernel_trace.csv> [step_index_from_end=2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the adam kernel (last launch of a step)
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
queues = {}
busy = []
for r in step:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]))
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}us q{q} wg={grid:5d} {'  ' * q}{name[:70]}")
    busy.append((s, e))
busy.sort()
idle, cur = 0.0, 0.0
for s, e in busy:
    if s > cur:
        idle += s - cur
    cur = max(cur, e)
print(f"step span {cur:.1f} us, {len(step)} launches, no-kernel-running time {idle:.1f} us")
