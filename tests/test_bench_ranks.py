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
@pytest.mark.parametrize("world", [2, 3])
def test_bench_band_path_runs_with_several_ranks_on_one_gpu(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HIKARI_BENCH_TRANSPORT="host", HIKARI_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "3", "--blocks", "2", "--width", "640", "--height", "360"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 6 and d["scaling"] == "strong" and d["config"]["parallelism"] == f"band{world}"
    assert d["value"] > 0 and d["rays_per_frame"] > 640 * 360 and "roofline" in d and d["config"]["halo_transport"] == "host"
    assert len(d["blocks_ms_per_step"]) == 2
