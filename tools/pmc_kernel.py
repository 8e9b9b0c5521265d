"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel name: mean counter value per dispatch.
  rocprofv3 --pmc C1 C2 ... --kernel-trace --output-format csv -d gpurun_out/x -- python bench.py ...
  python tools/pmc_kernel.py gpurun_out/x [substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub and sub not in k:
            continue
        a = acc[k[:90]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    n = max(v[0] for v in cs.values())
    print(f"{k}  ({n} dispatches)")
    for c, (m, v) in sorted(cs.items()):
        print(f"    {c:32s} {v / m:16.1f}")
