"""GPU box diagnostic (round 6): single-rank ladder / sharded ladder / sharded serial loop at a fixed PCG depth of 12 on the bench slice, pairwise differences, repeated."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as O
O.build()
import test_gpu_ladder as T
S = T.build_slice(O)
os.environ["I3D_EGT_TILE"] = "512"
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    os.environ.pop("I3D_LADDER", None)
    one = T._run(S, cg_fixed=12)
    lad = T._run_ranks(S, 2, cg_fixed=12)
    os.environ["I3D_LADDER"] = "1"
    ser = T._run_ranks(S, 2, cg_fixed=12)
    one_ser = T._run(S, cg_fixed=12)
    os.environ.pop("I3D_LADDER", None)
    print(f"rep {rep}: sdf  lad0-vs-lad1 {rel(lad[0][1], lad[1][1]):.2e}  ser0-vs-ser1 {rel(ser[0][1], ser[1][1]):.2e}  shardlad-vs-one {rel(lad[0][1], one[1]):.2e}  shardser-vs-one {rel(ser[0][1], one[1]):.2e}  "
          f"shardlad-vs-shardser {rel(lad[0][1], ser[0][1]):.2e}  one-ladder-vs-one-serial {rel(one[1], one_ser[1]):.2e}", flush=True)
    print("   stats one", [x[:3] for x in T._stats(one[0])], "\n   stats lad", [x[:3] for x in T._stats(lad[0][0])], "\n   stats ser", [x[:3] for x in T._stats(ser[0][0])], flush=True)
