"""Per-step memory-side traffic of the two-level round loop from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`bench.py --config two-level`: sums the dispatches of the LAST trace of each run (everything from its initStateKernel on).
Usage: python tools/pmc_two_level_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json> [rays]"""
import csv, glob, json, os, sys


def last_step(pass_dir, counter):
    f = sorted(glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    start = max(i for i, r in enumerate(rows) if "initStateKernel" in r["Kernel_Name"])
    per_kernel = {}
    for r in rows[start:]:
        k = r["Kernel_Name"]
        short = next((s for s in ("topTraverse", "bvh8TracePair", "bvh8Trace", "binByLeaf", "leafScan", "initState", "compact") if s in k), k[:32])
        per_kernel[short] = per_kernel.get(short, 0.0) + float(r["Counter_Value"])
    return per_kernel


fetch, write = last_step(sys.argv[1], "FETCH_SIZE"), last_step(sys.argv[2], "WRITE_SIZE")
kernels = sorted(set(fetch) | set(write))
# MI355X_MICROARCH.md, HBM section: both counters are KiB at the L2's memory side; on gfx950 FETCH_SIZE tallies the 128-byte
# requests of 16-byte-per-lane reads at 64 bytes, hence the factor 2 (an upper bound for kernels that read narrower)
by_kernel = {k: int((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024) for k in kernels}
out = {"traffic_bytes_per_step": sum(by_kernel.values()), "by_kernel": by_kernel, "fetch_size_kib": fetch, "write_size_kib": write,
       "rays_per_step": int(sys.argv[4]) if len(sys.argv) > 4 else None,
       "how": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over bench.py --config two-level, kernels of the last timed trace summed; (2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 half-count correction; includes Infinity Cache hits"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("traffic_bytes_per_step", "by_kernel")}, indent=1))
