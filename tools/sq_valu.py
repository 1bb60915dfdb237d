#!/usr/bin/env python3
"""VALU wave-instructions per launch of the two roofline kernels, for bench.py's `valu_frac` (what bounds k_build is instruction issue, not HBM).

    python tools/sq_valu.py <sq1.json from tools/pmc_summary.py> <bench json of the same command> <out: profiles/rNN_sq_counters.json>

SQ_INSTS_VALU is the mean over the dispatches that did work.  The fp64 share cannot be counted by a PMC counter on gfx950; it is the STATIC share of
fp64 VALU instructions in the kernel's ISA (hipcc -S of device/build.hip and device/tile_pass.hip for gfx950, counted by mnemonic), which for these
straight-line row loops is what the dynamic mix converges to: k_build<true> 885 fp64 and 421 packed-fp32 (v_pk_*) of 2804 VALU instructions, k_eg_tile 0
(fp32 only, a handful of fp64 conversions per tile).  Packed fp32 instructions are priced apart: 5.2 cycles per wave-instruction, like fp64
(tools/experiments/valu_rate.hip)."""
import json, re, sys
F64_SHARE = {"build": 885.0 / 2804.0, "cost": 799.0 / 1362.0, "eg_pass": 0.0}
PK_SHARE = {"build": 421.0 / 2804.0}
PAT = {"build": r"k_build<true", "cost": r"k_build<false", "eg_pass": r"k_eg_tile<", "eg_mr2": r"k_eg_tile_mr<2[,>]", "eg_mr3": r"k_eg_tile_mr<3[,>]", "observe": r"k_observe", "pcg_step": r"k_pcg_step3(_lad)?<1",
       "pcg_dir": r"k_pcg_dir3", "eg_gradcol": r"k_eg_gradcol"}
sq = json.load(open(sys.argv[1])); bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
out = {"source": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY over "
                 "`python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-kernel-timing` (no trace options in the same pass); means over the dispatches that did work",
       "kernel_tag": bench.get("kernel_tag"), "eg_rows": bench["config"]["rows"]["Eg"], "active_voxels": bench["config"]["active_voxels"], "kernels": {}}
for name, pat in PAT.items():
    for k, v in sq.items():
        if re.search(pat, k):
            m = v["mean"]; valu = m.get("SQ_INSTS_VALU", 0.0)
            e = {"kernel": k, "valu": valu, "valu_f64": valu * F64_SHARE.get(name, 0.0), "valu_pk": valu * PK_SHARE.get(name, 0.0), "f64_share_source": "static ISA mix (see tools/sq_valu.py)", "counters": m,
                 "valu_per_eg_row": valu * 64.0 / out["eg_rows"] if name in ("build", "cost", "eg_pass", "eg_mr2", "eg_mr3") else None}
            out["kernels"][name] = e
            print(name, k[-40:], "SQ_INSTS_VALU %.4g" % valu, "per Eg row (x64 lanes): %.0f" % (valu * 64.0 / out["eg_rows"]))
json.dump(out, open(sys.argv[3], "w"), indent=1)
