"""Summarises rocprofv3 --pmc CSV passes: per kernel, mean counter value per dispatch.
Usage: python tools/pmc_summary.py <dir with p1..pN pass directories> [<out pmc_traffic.json>]"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def issue_figures(c, n_xcd=8, n_simd=1024):
    """What bounds a kernel that is nowhere near the HBM roof, from the SQ / GRBM / TCC passes (c: counter name -> value per
    launch): VALU issue slots used (a wave64 VALU instruction holds its SIMD for 4 cycles), VALU lanes active, waves waiting,
    L2 hit rate, LDS bank-conflict share."""
    out = {}
    cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / n_xcd  # (the counter is summed over the XCDs)
    if cycles and "SQ_INSTS_VALU" in c:
        out["valu_insts"] = c["SQ_INSTS_VALU"]
        out["gpu_cycles"] = cycles
        out["valu_issue_frac"] = 4.0 * c["SQ_INSTS_VALU"] / (n_simd * cycles)
    if c.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in c:
        out["valu_active_lane_frac"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
        out["wave_wait_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if c.get("TCC_REQ_sum") and "TCC_HIT_sum" in c:
        out["l2_hit_frac"] = c["TCC_HIT_sum"] / c["TCC_REQ_sum"]
    out["how"] = ("rocprofv3 --pmc SQ / GRBM / TCC passes of the same command (tools/pmc_passes.sh): valu_issue_frac = 4 x SQ_INSTS_VALU / "
                  "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) -- a wave64 VALU instruction holds its SIMD for 4 cycles; packed and DPP forms count once, so "
                  "values slightly above 1 occur; valu_active_lane_frac = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); wave_wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
    return out


def read_passes(root):
    """kernel (short name) -> counter -> [(dispatch id, value)] over the newest run of every pass directory"""
    acc = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, "p*"))):
        if not os.path.isdir(d):
            continue
        found = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not found:
            continue
        with open(found[-1]) as fh:  # a pass directory that was reused keeps older runs: only the newest counts
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if "rocclr" in k:
                    continue
                m = re.search(r"(\w+Kernel)", k)
                short = m.group(1) if m else k[:40]
                acc[short][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return acc


def mean_tail(series):
    vals = [v for _, v in sorted(series)]
    tail = vals[1:] if len(vals) > 1 else vals  # first dispatch of bench.py is the primary-ray pass
    return sum(tail) / len(tail)


def main():
    root = sys.argv[1]
    acc = read_passes(root)
    for kern, ctrs in acc.items():
        print(kern)
        for name in sorted(ctrs):
            vals = [v for _, v in sorted(ctrs[name])]
            print("  %-32s n=%d mean(all but first)=%.6g  last=%.6g  sum=%.6g" % (name, len(vals), mean_tail(ctrs[name]), vals[-1], sum(vals)))
    # HBM-side traffic of the dominant kernel, per launch (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are
    # KiB at the L2's memory side; on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64
    # bytes, so it is doubled; WRITE_SIZE is exact). Infinity-Cache hits are included in both.
    if len(sys.argv) > 2 and "bvh8TraceKernel" in acc:
        c = acc["bvh8TraceKernel"]

        def rays_per_launch():
            for lf in sorted(glob.glob(os.path.join(root, "p*.log"))):
                m = re.search(r'"rays_per_gpu": (\d+)', open(lf, errors="replace").read())
                if m:
                    return int(m.group(1))
            return None

        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = mean_tail(c["FETCH_SIZE"]), mean_tail(c["WRITE_SIZE"])
            out = {"fetch_size_kib": f, "write_size_kib": w, "traffic_bytes_per_launch": int((2.0 * f + w) * 1024),
                   "rays_per_launch": rays_per_launch(),
                   "how": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over bench.py, mean over the timed launches; (2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 half-count correction for 16-byte-per-lane reads; includes Infinity Cache hits",
                   "issue": issue_figures({k: mean_tail(v) for k, v in c.items()})}
            with open(sys.argv[2], "w") as fh:
                json.dump(out, fh, indent=1)
            print("traffic json ->", sys.argv[2], out["traffic_bytes_per_launch"], json.dumps(out["issue"]))


if __name__ == "__main__":
    main()
