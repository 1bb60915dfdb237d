"""Where a kernel waits for ONE load at a time: for every kernel of a `-save-temps` device assembly (.s), the `s_waitcnt vmcnt(0)` that follow fewer than N global / buffer loads since the
previous vmcnt wait — each is a round trip the wave sits through with nothing else of its own in flight (round 6: k_build<true> had 34 of them, 18 in a row in its regulariser
set-up).  usage: python tools/isa_drains.py <file.s> [max_loads_in_front=2]"""
import re, sys
path = sys.argv[1]; lim = int(sys.argv[2]) if len(sys.argv) > 2 else 2
name = None; stats = {}
for ln in open(path):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1); stats[name] = {"loads": 0, "drains": 0, "lonely": 0, "since": 0, "lines": 0}; continue
    if name is None: continue
    s = stats[name]; s["lines"] += 1
    if re.search(r"\b(global_load|buffer_load|flat_load)", ln): s["loads"] += 1; s["since"] += 1
    elif re.search(r"s_waitcnt.*vmcnt\(0\)", ln):
        s["drains"] += 1
        if 0 < s["since"] <= lim: s["lonely"] += 1
        s["since"] = 0
    elif re.search(r"s_waitcnt.*vmcnt\(\d+\)", ln): s["since"] = 0
    if ln.startswith(".Lfunc_end"): name = None          # (a kernel has several s_endpgm: early exits)
import subprocess
def demangle(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()[:90]
    except Exception: return n[:90]
print("%-92s %6s %6s %8s %8s" % ("kernel", "lines", "loads", "vmcnt(0)", "lonely"))
for n, s in sorted(stats.items(), key=lambda kv: -kv[1]["lonely"]):
    if s["loads"]: print("%-92s %6d %6d %8d %8d" % (demangle(n), s["lines"], s["loads"], s["drains"], s["lonely"]))
