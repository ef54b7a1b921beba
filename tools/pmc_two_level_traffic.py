"""Per-step memory-side traffic and issue figures of the two-level round loop from rocprofv3 --pmc passes over
`bench.py --config two-level` (tools/pmc_passes.sh <tag> --config two-level): sums the dispatches of the LAST trace of each
run, per kernel: everything from the first initStateKernel of that trace on -- a trace of a large batch runs as PARTS batches side by
side (pgpu_set_batch_parts, default 2), each with an initStateKernel of its own.
Usage: [PARTS=2] python tools/pmc_two_level_traffic.py <dir with p1..pN pass directories> <out.json> [rays]"""
import csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import issue_figures

NAMES = ("topTraverse", "fusedTrace", "bvh8TracePair", "bvh8TraceInst", "bvh8Trace", "binByLeaf", "leafScan", "initState", "compact")


def last_step(pass_dir):
    """counter -> kernel -> sum over the last step"""
    found = sorted(glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not found:
        return {}
    rows = list(csv.DictReader(open(found[-1])))
    out = {}
    for counter in sorted({r["Counter_Name"] for r in rows}):
        rs = sorted((r for r in rows if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
        inits = [i for i, r in enumerate(rs) if "initStateKernel" in r["Kernel_Name"]]
        start = inits[-min(len(inits), int(os.environ.get("PARTS", "2")))]
        per = {}
        for r in rs[start:]:
            k = r["Kernel_Name"]
            short = next((s for s in NAMES if s in k), k[:32])
            per[short] = per.get(short, 0.0) + float(r["Counter_Value"])
        out[counter] = per
    return out


def main():
    root = sys.argv[1]
    counters = {}
    for d in sorted(glob.glob(os.path.join(root, "p*"))):
        if os.path.isdir(d):
            counters.update(last_step(d))
    fetch, write = counters.get("FETCH_SIZE", {}), counters.get("WRITE_SIZE", {})
    kernels = sorted(set(fetch) | set(write))
    # MI355X_MICROARCH.md, HBM section: both counters are KiB at the L2's memory side; on gfx950 FETCH_SIZE tallies the 128-byte
    # requests of 16-byte-per-lane reads at 64 bytes, hence the factor 2 (an upper bound for kernels that read narrower)
    by_kernel = {k: int((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024) for k in kernels}
    issue = {}
    for k in kernels:
        c = {name: per[k] for name, per in counters.items() if k in per}
        fig = issue_figures(c)
        how = fig.pop("how")
        if len(fig) > 0 and c.get("SQ_INSTS_VALU", 0) > 1e6:
            issue[k] = fig
    issue["how"] = how if kernels else None
    out = {"traffic_bytes_per_step": sum(by_kernel.values()), "by_kernel": by_kernel, "fetch_size_kib": fetch, "write_size_kib": write,
           "rays_per_step": int(sys.argv[3]) if len(sys.argv) > 3 else None, "issue": issue,
           "how": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over bench.py --config two-level, kernels of the last timed trace summed; (2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 half-count correction; includes Infinity Cache hits"}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("traffic_bytes_per_step", "by_kernel", "issue")}, indent=1))


if __name__ == "__main__":
    main()
