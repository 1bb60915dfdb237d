"""GPU box: the launch sequence of ONE outer iteration from a rocprofv3 kernel trace (CSV), run-length encoded, with the time each name takes and the idle time in front of it.
usage: python tools/experiments/iteration_sequence.py <kernel_trace.csv> [which_iteration]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n); n = n.replace("i3d::", "")
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    m = re.match(r"rocprim::trampoline_kernel<rocprim::([a-z_A-Z0-9]+)", n)
    if m: n = "rocprim:" + m.group(1)
    return n[:60]
starts = [i for i, r in enumerate(rows) if "k_classify" in r["Kernel_Name"]]
a, b = starts[which], starts[which + 1]
seq = rows[a:b]
t0 = int(seq[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
print("iteration %d: %d launches, %.3f ms wall" % (which, len(seq), (t1 - t0) / 1e6))
tot = {}; prev_end = None; out = []
for r in seq:
    n = short(r["Kernel_Name"]); s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    gap = 0 if prev_end is None else max(0, s - prev_end)
    d = tot.setdefault(n, [0, 0, 0]); d[0] += 1; d[1] += e - s; d[2] += gap
    prev_end = max(e, prev_end or 0)
    out.append(n)
print("%-62s %6s %10s %10s" % ("kernel", "calls", "busy_us", "gap_us_in_front"))
for n, d in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %6d %10.1f %10.1f" % (n, d[0], d[1] / 1e3, d[2] / 1e3))
print("busy %.3f ms, gaps %.3f ms" % (sum(d[1] for d in tot.values()) / 1e6, sum(d[2] for d in tot.values()) / 1e6))
# run-length encoded sequence
rle = []; 
for n in out:
    if rle and rle[-1][0] == n: rle[-1][1] += 1
    else: rle.append([n, 1])
print(" | ".join(n if k == 1 else "%s x%d" % (n, k) for n, k in rle))
