#!/usr/bin/env python3
"""Average duration per kernel from a rocprofv3 --kernel-trace CSV, with and without the launches that returned at once
(PCG launches queued behind the device-side convergence flag last ~4 us; bench.py excludes them the same way: a launch counts when it
lasted >= 25 % of the longest launch of that kernel).

    python tools/kernel_trace_avg.py gpurun_out/prof/<run>/<pid>_kernel_trace.csv [regex]
"""
import collections, csv, re, sys
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    if pat and not pat.search(r["Kernel_Name"]):
        continue
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>9s} | {'work':>6s} {'avg_us':>9s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    mx = max(v); w = [x for x in v if x >= 0.25 * mx]
    print(f"{k[-60:]:60s} {len(v):6d} {sum(v) / len(v):9.1f} | {len(w):6d} {sum(w) / len(w):9.1f}")
