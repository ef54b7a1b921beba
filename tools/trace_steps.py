"""Per-kernel time per step from a rocprofv3 --kernel-trace CSV of bench.py --config two-level (last step of the run)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("topTraverse", "svdagCast", "svdagPlan", "leafScan", "binByLeaf", "bvh8TracePair", "bvh8TraceInst", "bvh8Trace", "initState", "fillBuffer", "copyBuffer"):
        if k in n:
            return k
    return n[:30]
ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a step starts with initState
starts = [i for i, e in enumerate(ev) if e[0] == "initState"]
s = starts[-1]
step = ev[s:]
# cut at the long copy tail (cpu check) if any: stop at first gap > 1 ms
out = [step[0]]
for e in step[1:]:
    if e[1] - out[-1][2] > 1_000_000:
        break
    out.append(e)
tot = collections.OrderedDict()
for n, a, b in out:
    tot[n] = tot.get(n, 0) + (b - a) / 1e3
busy = sum(tot.values())
span = (out[-1][2] - out[0][1]) / 1e3
print("kernels in step:", len(out), "span %.1f us, busy %.1f us, idle %.1f us" % (span, busy, span - busy))
for k, v in tot.items():
    print("  %-16s %9.1f us" % (k, v))
if len(sys.argv) > 2:
    for n, a, b in out:
        print("%-16s %9.1f  gap_before %7.1f" % (n, (b - a) / 1e3, 0.0))
