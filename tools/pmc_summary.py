#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (optionally filtered by regex).  Launches that returned
at once (PCG launches queued behind the device-side convergence flag) are dropped: a dispatch counts when its SQ_WAVE_CYCLES (or, without
that counter, its first counter) reaches 25 % of the kernel's maximum.

    python tools/pmc_summary.py <dir> [kernel regex] [--json out.json]
"""
import collections, csv, glob, json, re, sys
args = [a for a in sys.argv[1:] if not a.startswith("--json")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out_json: args = [a for a in args if a != out_json]
pat = re.compile(args[1]) if len(args) > 1 else None
disp = collections.defaultdict(lambda: collections.defaultdict(dict))          # kernel -> dispatch -> counter -> value
for f in glob.glob(args[0] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat and not pat.search(row["Kernel_Name"]): continue
        k = row["Kernel_Name"].split("(")[0][-60:]
        d = disp[k][(f, row["Dispatch_Id"])]
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
res = {}
for k, ds in disp.items():
    ref = "SQ_WAVE_CYCLES" if any("SQ_WAVE_CYCLES" in d for d in ds.values()) else sorted(next(iter(ds.values())).keys())[0]
    mx = max(d.get(ref, 0.0) for d in ds.values())
    work = [d for d in ds.values() if d.get(ref, 0.0) >= 0.25 * mx]
    print(f"{k}   ({len(work)} of {len(ds)} dispatches did work)")
    res[k] = {"dispatches": len(ds), "work_dispatches": len(work), "mean": {}}
    for c in sorted({c for d in work for c in d}):
        v = [d[c] for d in work if c in d]
        res[k]["mean"][c] = sum(v) / len(v)
        print(f"   {c:34s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
