#!/usr/bin/env python3
"""Device idle of the timed region from a rocprofv3 --kernel-trace of bench.py: wall span of the last 10 Gauss-Newton iterations against the sum of the
durations of ALL kernels in it (the library's, rocPRIM's sorts / scans, copies done by kernels).  usage: tools/timeline_idle.py <kernel_trace.csv>"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
builds = [k for k in ks if "k_build<true" in k[2]]
n_it = min(10, len(builds))
t0 = [k for k in ks if "k_classify" in k[2] and k[0] < builds[-n_it][0]][-1][0]      # the assembly of that iteration starts with the classification
t1 = [k for k in ks if "k_lm_decide" in k[2] or "k_accept" in k[2]][-1][1]      # the region ends with the last LM attempt (the bench's read-back of the grid follows)
sel = [k for k in ks if k[0] >= t0 and k[1] <= t1]
span = sel[-1][1] - sel[0][0]; busy = sum(e - s for s, e, _ in sel)
gaps = [max(0, sel[i + 1][0] - sel[i][1]) for i in range(len(sel) - 1)]
print("last %d iterations: span %.2f ms, kernels %.2f ms in %d launches -> %.3f ms per iteration of wall clock, %.3f ms of it idle; gap between consecutive kernels: median %.2f us, mean %.2f us; "
      "%d gaps above 20 us hold %.2f ms" % (n_it, span / 1e6, busy / 1e6, len(sel), span / 1e6 / n_it, (span - busy) / 1e6 / n_it, statistics.median(gaps) / 1e3, statistics.mean(gaps) / 1e3,
                                             sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
big = sorted(((g, sel[i][2][:48], sel[i + 1][2][:48]) for i, g in enumerate(gaps)), reverse=True)[:10]
for g, a, b in big:
    print("%8.1f us  after %-50s before %s" % (g / 1e3, a, b))
