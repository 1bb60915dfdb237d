#!/usr/bin/env python3
"""Instruction / load / s_waitcnt statistics per kernel of a device file's ISA (hipcc -S --cuda-device-only):
   tools/isa_stats.py device/tile_pass.hip [name filter]     -- run from intrinsic3d_amd/csrc (no GPU needed)"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:] 
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-command-line-argument",
                "-S", "--cuda-device-only", "-o", out, src] + extra, check=True)
txt = open(out).read()
for m in re.finditer(r'\n(_Z[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end', txt, re.S):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name)
    if flt and flt not in name: continue
    ins = [l.strip() for l in m.group(2).split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    vm = [l for l in ins if l.startswith('s_waitcnt') and 'vmcnt' in l]
    print("%-62s %6d instr  %4d vmem loads  %3d vmcnt waits (%d vmcnt(0))  %3d scratch  %3d ds_add_f32  %4d s_barrier" % (
        name[:62], len(ins), sum(1 for l in ins if re.match(r'(buffer|global|flat)_load', l)), len(vm), sum(1 for l in vm if 'vmcnt(0)' in l),
        sum(1 for l in ins if l.startswith('scratch_')), sum(1 for l in ins if l.startswith('ds_add_f32') or l.startswith('ds_add_rtn_f32')), sum(1 for l in ins if l.startswith('s_barrier'))))
print("ISA:", out)
