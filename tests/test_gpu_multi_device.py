"""Checks that need MORE THAN ONE device: they start themselves when the box has them (the driver's 8-GPU node) and skip on a 1-GPU box.
One process per GPU, the library's RCCL communicator, the mailbox transport decided by its start-up test between real devices."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _launch(world, args, timeout=900, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("transport", ["p2p", "rccl"])
def test_sharded_processes_match_single_rank(tmp_path, transport):
    """tests/multi_rank_worker.py on 1 device (plain path) and on every power-of-two device count the box offers: rows, accept / reject sequence, costs, fields and
    camera of EVERY rank equal the single-rank run's to the sharded path's bar; over RCCL (the default: the six-launch pass) and over the mailboxes
    (I3D_TRANSPORT=p2p, when their start-up test passes between the devices: three launches per PCG pass, the exchanges inside the kernels)."""
    nd = _devices()
    if nd < 2:
        pytest.skip("needs at least two devices")
    worker = os.path.join(ROOT, "tests", "multi_rank_worker.py")
    base = str(tmp_path / "ref")
    r = _launch(1, [worker, base, "2", "12"])
    assert r.returncode == 0, r.stdout + r.stderr
    ref = np.load(base + ".rank0.npz")
    worlds = [w for w in (2, 4, 8) if w <= nd]
    took_over = False
    for W in worlds:
        out = str(tmp_path / f"w{W}")
        r = _launch(W, [worker, out, "2", "12"], env_extra={"I3D_TRANSPORT": transport})
        assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
        for k in range(W):
            d = np.load(out + f".rank{k}.npz")
            # the mailboxes carry the exchanges only when their start-up test passed on every rank: on a node where it does not, RCCL stays in charge and the run must still match
            assert str(d["transport"]).startswith("rccl") or (transport == "p2p" and str(d["transport"]).startswith("p2p-mailbox")), str(d["transport"])
            print(f"[multi-device] W = {W}, asked for {transport}: {d['transport']}")
            took_over = took_over or (transport == "p2p" and str(d["transport"]).startswith("rccl"))
            assert np.array_equal(d["rows"], ref["rows"]) and np.array_equal(d["accepted"], ref["accepted"])
            assert np.all(np.abs(d["cost"] - ref["cost"]) <= 1e-4 * np.abs(ref["cost"]))
            assert np.abs(d["sdf"] - ref["sdf"]).max() <= 1e-4 * np.abs(ref["sdf"]).max() and np.abs(d["alb"] - ref["alb"]).max() <= 1e-4 * np.abs(ref["alb"]).max()
            np.testing.assert_allclose(d["intr"], ref["intr"], rtol=1e-4); np.testing.assert_allclose(d["poses"], ref["poses"], rtol=1e-4, atol=1e-6)
            assert int(d["halo_calls"]) > 0 and int(d["reduce_calls"]) > 0
    if took_over:      # advisor finding of round 5: a mailbox bootstrap that fails must not read as "the p2p variant passed" — the run above re-tested RCCL
        pytest.skip("I3D_TRANSPORT=p2p was asked for but the mailbox start-up test did not pass between these devices: RCCL carried the run (results matched); the mailbox transport stays unvalidated here")


def test_bench_scales_over_the_devices_of_the_box(tmp_path):
    """bench.py --gpus N on a small problem for N = 1 and the largest power of two the box offers: it starts its ranks itself, every run prints ONE line, the sharded
    line carries the transport, and the row counts of the two runs agree (same problem, strong scaling)."""
    nd = _devices()
    if nd < 2:
        pytest.skip("needs at least two devices")
    W = max(w for w in (2, 4, 8) if w <= nd)
    lines = {}
    for n in (1, W):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--voxels", "1e6", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--band2-steps", "0"],
                           capture_output=True, text=True, timeout=1200, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(out) == 1, r.stdout
        lines[n] = json.loads(out[0])
    assert lines[W]["n_gpus"] == W and lines[1]["n_gpus"] == 1
    assert lines[W]["config"]["rows"] == lines[1]["config"]["rows"]
    assert lines[W]["comm"] and lines[W]["comm"]["transport"]
    (tmp_path / "scale.json").write_text(json.dumps(lines))
