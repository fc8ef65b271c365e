"""GPU suite: bench.py's multi-rank path exactly as the driver launches it,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

as a fresh child process (the launcher starts first; nothing in it has touched the GPU).  A gpurun box has ONE MI355X, so
the ranks share cuda:0 and RHJ_BENCH_BACKEND=gloo moves the exchange through the host -- everything else (sharded
schedule, barriers, max-over-ranks timing, verification by all-reduced count + checksum, the JSON line) is the code an
8-GPU RCCL run executes.  Config 5 of BASELINE.json at reduced size, uniform and Zipf."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("ranks,dist_,tuples", [(2, "uniform", 6_000_000), (3, "zipf", 4_000_000), (2, "zipf", 3_000_000)])
def test_bench_multirank_gloo_rehearsal(ranks, dist_, tuples, tmp_path):
    env = dict(os.environ, RHJ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--tuples", str(tuples), "--dist", dist_, "--cpu-sample", "0", "--no-extras"]
    err = open(tmp_path / "stderr.txt", "wb")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=err, timeout=900)
    err.close()
    tail = open(tmp_path / "stderr.txt", "rb").read()[-3000:].decode(errors="replace")
    assert r.returncode == 0, tail
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                 # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["verified"] is True
    assert line["n_gpus"] == ranks and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["unit"] == "tuples/s" and line["value"] > 0
    assert line["config"]["tuples_R_global"] == ranks * tuples
    # the transport actually used is named (a gloo rehearsal must not claim RCCL)
    assert "gloo" in line["config"]["exchange"] and "RCCL" not in line["config"]["exchange"]
    sh = line["sharded"]
    assert sh["backend"] == "gloo" and sh["wire_format"] == "narrow12" and sh["bytes_per_tuple_sent"] == 12
    assert sh["local_plan"] == [2, 8, 8]
    # 12 B per tuple that leaves the rank: both relations, minus what the rank keeps (about 1 / ranks of them)
    sent_tuples = sh["exchange_bytes_per_rank"] / 12
    assert sh["exchange_bytes_per_rank"] % 12 == 0
    assert 0.5 * 2 * tuples * (ranks - 1) / ranks <= sent_tuples <= 2 * tuples
    # balanced class ranges: rank 0 receives about its share even with a Zipf foreign key
    assert sum(sh["recv_tuples_rank0"]) <= 1.3 * 2 * tuples
    assert sh["kernel_ms_per_step_rank0"]["join"] > 0 and sh["kernel_ms_per_step_rank0"]["scatter"] > 0
    assert sh["ranks_seen"] == ranks and sh["ranks_in_all_reduce"] == ranks      # the collectives really spanned N ranks


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment (how a driver without torchrun calls
    it): the script starts torch.distributed.run as a child before touching the GPU and relays rank 0's line and the rc"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RHJ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tuples", "3000000",
           "--cpu-sample", "0", "--no-extras"]
    err = open(tmp_path / "stderr.txt", "wb")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=err, timeout=900)
    err.close()
    assert r.returncode == 0, open(tmp_path / "stderr.txt", "rb").read()[-3000:].decode(errors="replace")
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["verified"] is True and line["n_gpus"] == 2 and line["sharded"]["ranks_seen"] == 2
    # a failing child's exit code comes back: an impossible plan makes every rank exit non-zero
    bad = subprocess.run(cmd + ["--bits1", "99"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
    assert bad.returncode != 0
