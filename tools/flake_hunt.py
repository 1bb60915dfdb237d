#!/usr/bin/env python3
"""Dev tool (GPU box): which stage of one outer iteration differs between two runs on identical inputs?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, test_gpu_bench_parity as T
from oracle import oracle_py as O
O.build(); S = T.build_slice(O)
sc = S["sc"]; a0 = S["arrays"]
first = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    ocfg = T._bench_cfg(O, S["thres"], int(sys.argv[2]) if len(sys.argv) > 2 else -1); ocfg.iterations = 1
    cfg = helpers.gpu_cfg(ocfg)
    ctx = helpers.gpu_context(sc, a0, S["vsh"])
    ctx.debug_assemble(cfg, 0)
    fl = ctx.debug_flags()
    f, w, res, J = ctx.debug_eg_rows(jac=True); v = f
    g, dg, cost = ctx.debug_normal_eq()
    rng = np.random.default_rng(0); x = rng.normal(0, 1, g.shape)
    y = ctx.debug_jtj_apply(x)
    st = ctx.optimize(cfg)
    sdf, alb = ctx.get_grid(); gi, gd, gp = ctx.get_camera()
    ctx.close()
    cur = dict(fl=fl, v=v, f=f, w=w, res=res, J=J, g=g, dg=dg, cost=np.array([cost]), y=y, sdf=sdf, alb=alb, poses=gp, intr=gi)
    if first is None:
        first = cur; continue
    out = []
    for k in cur:
        a, b = np.asarray(cur[k], np.float64), np.asarray(first[k], np.float64)
        if a.shape != b.shape: out.append(f"{k}: SHAPE {a.shape} vs {b.shape}"); continue
        d = np.abs(a - b).max() if a.size else 0.0
        out.append(f"{k} {d:.2e}")
    print(f"rep {rep}: " + "  ".join(out) + f"   pcg {list(st[0].pcg_iterations[:st[0].num_attempts])}", flush=True)
