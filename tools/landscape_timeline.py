"""Where a step of the streamed landscape goes: from a rocprofv3 --kernel-trace csv (tools/landscape_profile.sh), the last trace
(from its last initStateKernel on) split into time the GPU traverses, time it copies (gather kernels), both at once, neither.
Usage: python tools/landscape_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start = max(i for i, r in enumerate(rows) if "initStateKernel" in r["Kernel_Name"])
rows = rows[start:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)


def short(name):
    for k in ("topTraverse", "bvh8TraceInst", "bvh8TracePair", "bvh8Trace", "binByLeaf", "leafScan", "leafGather", "gather", "splitByLeaf", "leafTable", "leafToBvh", "addCounts", "initState", "fillBuffer", "copyBuffer"):
        if k in name:
            return k
    return name[:40]


def union(iv):
    iv = sorted(iv)
    out, cur = [], None
    for a, b in iv:
        if cur is None or a > cur[1]:
            if cur:
                out.append(cur)
            cur = [a, b]
        else:
            cur[1] = max(cur[1], b)
    if cur:
        out.append(cur)
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    out, i, j = [], 0, 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


per = {}
for r in rows:
    k = short(r["Kernel_Name"])
    d = per.setdefault(k, [0, 0])
    d[0] += 1
    d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
is_copy = lambda r: "ather" in r["Kernel_Name"]
copy = union([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if is_copy(r)])
comp = union([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if not is_copy(r)])
both = intersect(copy, comp)
busy = union([tuple(x) for x in copy] + [tuple(x) for x in comp])
ms = 1e-6
print("step %.2f ms: traversal kernels busy %.2f ms, gather kernels busy %.2f ms, both at once %.2f ms, GPU idle %.2f ms" % (
    (t1 - t0) * ms, length(comp) * ms, length(copy) * ms, length(both) * ms, ((t1 - t0) - length(busy)) * ms))
for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-16s %5d launches %8.2f ms" % (k, n, d * ms))
# the step in tenths: what runs when
for q in range(10):
    a, b = t0 + (t1 - t0) * q // 10, t0 + (t1 - t0) * (q + 1) // 10
    w = [[a, b]]
    print("  tenth %d: traversal %.2f ms, copies %.2f ms, idle %.2f ms" % (q, length(intersect(comp, w)) * ms, length(intersect(copy, w)) * ms,
                                                                       ((b - a) - length(intersect(busy, w))) * ms))
