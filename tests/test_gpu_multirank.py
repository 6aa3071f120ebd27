"""The N > 1 path on real hardware, rehearsed on the one GPU a test box has: two ranks (gloo for the barrier
and the two scalar reductions, both ranks on cuda:0) run bench.py's sharded step -- block b of the global
batch goes to rank b mod 2, no data-path collective -- and must produce, summed over ranks, exactly the coded
bytes one rank produces for the same 512 global blocks (position-stable, bit-identical for any G)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_two_rank_rehearsal_matches_one_rank(gpu_ctx):
    common = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--state-budget-gib", "24"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "512"] + common,
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    r1 = _json_line(one.stdout)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                          "--blocks", "256"] + common,
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    r2 = _json_line(two.stdout)
    assert r1["n_gpus"] == 1 and r2["n_gpus"] == 2
    assert r1["roundtrip_bit_exact"] and r2["roundtrip_bit_exact"]
    assert r1["config"]["blocks_per_gpu"] == 512 and r2["config"]["blocks_per_gpu"] == 256
    assert r1["coded_bytes"] == r2["coded_bytes"] > 0          # same 512 global blocks, whoever coded them
    assert r2["scaling"] == "weak" and r2["value"] > 0


def test_plain_invocation_starts_its_own_ranks(gpu_ctx):
    """`python bench.py --gpus 2 ...` with no launcher on the command line (the shape of the driver's N = 1 command)
    must BE a 2-rank run: it starts the ranks itself, labels the line n_gpus = 2 and lists one device entry per rank."""
    common = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--state-budget-gib", "24"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "512"] + common,
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    # no --backend: the DEFAULT process group (gloo: CPU tensors for the barrier and the two reductions) is what the driver's
    # N > 1 command line gets, so this is the path a SCALE run takes ("what's weak" 9)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device",
                          "--blocks", "256"] + common, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    r1, r2 = _json_line(one.stdout), _json_line(two.stdout)
    assert r2["n_gpus"] == 2 and len(r2["devices"]) == 2 and r2["backend"] == "gloo"
    assert sorted(d["rank"] for d in r2["devices"]) == [0, 1]
    assert len({d["pid"] for d in r2["devices"]}) == 2           # two processes really ran
    assert r2["roundtrip_bit_exact"] and r1["coded_bytes"] == r2["coded_bytes"] > 0
