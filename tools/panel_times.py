"""Per-dispatch durations of the fused row-panel kernel from a rocprofv3 --kernel-trace CSV, grouped by grid size and
rounded duration: separates the per-slot cost from the fixed per-panel cost (pre: 36 slots, layers: 144 / 168, last: 108)."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dit_panel" in r["Kernel_Name"]:
            rows.append((int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
g = defaultdict(list)
for grid, us in rows:
    g[(grid, round(us / 10) * 10)].append(us)
for k in sorted(g):
    v = g[k]
    print(f"grid {k[0]:>8}  ~{k[1]:>5} us : n={len(v):4d}  mean {sum(v)/len(v):7.1f} us  min {min(v):7.1f}")
