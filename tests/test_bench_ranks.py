"""bench.py's multi-rank path (one process per rank, band-sharded frame, halo exchanges, max-over-ranks timing, summed ray
counts) on ONE GPU: two and three ranks share device 0 and exchange halos over gloo (RCCL refuses two ranks on one device).
The numbers mean nothing - the point is that every line the driver's N = 2, 4, 8 runs execute has run before."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("world,launcher", [(2, "torchrun"), (3, "torchrun"), (2, "plain"), (3, "balanced"), (3, "rebalanced")])
def test_bench_band_path_runs_with_several_ranks_on_one_gpu(world, launcher):
    """launcher "torchrun": the driver's own command line.  "plain": `python bench.py --gpus N` with no launcher - bench.py
    re-launches itself under torch.distributed.run (VERDICT r02 missing 1)."""
    env = dict(os.environ, HIKARI_BENCH_TRANSPORT="host", HIKARI_BENCH_DEVICE="0")
    for attempt in range(3):   # (the launcher needs a TCP port; one picked by bind / close can be taken before the launcher binds it)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable]
        if launcher in ("torchrun", "balanced", "rebalanced"):
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port)]
        cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "3", "--blocks", "2", "--width", "640", "--height", "360"]
        if launcher == "balanced":   # the split by cost, derived by every rank on frame 1 (HK_FRAME_BALANCE_BANDS / hk_balance_bands)
            cmd += ["--band-split", "balanced"]
        if launcher == "rebalanced":   # round 6: the split follows measured band times during a longer warm-up (all-gather of N floats, migration of the moved rows)
            cmd[cmd.index("--warmup") + 1] = "24"
            cmd += ["--band-rebalance-rounds", "4"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode == 0 or "address already in use" not in r.stderr.lower():
            break
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 alone prints
    d = json.loads(lines[0])
    if launcher == "balanced":
        b = d["config"]["band_bounds"]
        assert d["config"]["band_split"] == "balanced" and len(b) == world + 1 and b[0] == 0 and b[-1] == 360 and d["replay_bit_identical"], d["config"]
        assert b != [0, 120, 240, 360]   # the Cornell box sits in the middle rows: the middle band is thinner
    elif launcher == "rebalanced":
        rb = d["config"]["band_rebalancing"]
        assert d["config"]["band_split"] == "measured" and rb["rounds_during_warmup"] == 4 and len(rb["splits_taken"]) >= 1, d["config"]
        b = d["config"]["band_bounds"]
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == 360 and d["replay_bit_identical"]   # (the replays follow the same splits: migration included)
    else:
        assert d["config"]["band_split"] == "equal"
    assert d["n_gpus"] == world and d["steps"] == 6 and d["scaling"] == "strong" and d["config"]["parallelism"] == f"band{world}"
    assert d["value"] > 0 and d["rays_per_frame"] > 640 * 360 and "roofline" in d and d["config"]["halo_transport"] == "host"
    assert len(d["blocks_ms_per_step"]) == 2
    assert d["config"]["gather"].startswith("rank 0 collects") and d["replay_bit_identical"]   # (SURVEY 8e step 7 runs inside the timed frames)


@pytest.mark.gpu
def test_bench_fails_instead_of_falling_back_when_rccl_cannot_come_up():
    """Two ranks on ONE device: RCCL refuses ("duplicate GPU").  bench.py must exit non-zero on every rank - an N-GPU line is
    never a host-staged number in disguise - and print no JSON line."""
    env = dict(os.environ, HIKARI_BENCH_DEVICE="0")
    env.pop("HIKARI_BENCH_TRANSPORT", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "1", "--width", "320", "--height", "180",
           "--no-hbm-probe"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "RCCL halo transport did not come up" in r.stderr or "ncclCommInitRank" in r.stderr, r.stderr[-3000:]


def test_bench_refuses_more_ranks_than_gpus_without_a_launcher():
    """CPU container: no GPU at all -> a loud exit, not a CPU fallback."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and ("needs an MI355X" in r.stderr or "exposes" in r.stderr)
