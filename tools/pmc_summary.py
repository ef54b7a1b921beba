"""Summarises rocprofv3 --pmc CSV passes: per kernel, mean counter value per dispatch."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "rocclr" in k:
                continue
            short = k.split("(")[0].split("::")[-1]
            acc[short][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for kern, ctrs in acc.items():
    print(kern)
    for name in sorted(ctrs):
        vals = [v for _, v in sorted(ctrs[name])]
        tail = vals[1:] if len(vals) > 1 else vals  # first dispatch of bench.py is the primary-ray pass
        print("  %-32s n=%d mean(all but first)=%.6g  last=%.6g" % (name, len(vals), sum(tail) / len(tail), vals[-1]))
