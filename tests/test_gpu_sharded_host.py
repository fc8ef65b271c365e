"""GPU suite: the multi-GPU join from a plain C++ host (radixhashjoin_amd/host/sharded_host.cpp): RCCL collectives
(ncclAllGather, grouped ncclSend / ncclRecv, ncclAllReduce) + the rhj_shard_* stage calls of include/rhj.h, no Python and no
torch.  A gpurun box has one GPU and RCCL wants one GPU per rank, so this is the ONE-RANK run of the program an 8-GPU node
would start eight times.  What executes, and what does not:
  * default run: ncclCommInitRank, ncclAllGather, ncclAllReduce and every C-ABI call; the rank keeps its own segment with a
    device copy, so the grouped ncclSend / ncclRecv loop does NOT run (there is no peer);
  * RHJ_SHARD_VIA_SELF=1 + RHJ_SHARD_MAX_MSG=<tuples>: the own segment goes through ncclSend / ncclRecv in pieces, as a peer's
    would -- the piece loop and RCCL's in-order matching of several messages to one peer execute (sizes the self-message
    probe found correct: <= 0.8 GB per message, profiles/r03_rccl_self_message_probe.txt).
More than one rank over RCCL is unverified on this pool; the peer schedule is covered by the gloo tests of sharded.py.
The program verifies its own pair set (count + checksum against the closed form, all-reduced)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "radixhashjoin_amd", "host", "sharded_host")


@pytest.mark.parametrize("rows,dist_", [(12_000_000, "uniform"), (20_000_000, "zipf"),
                                        (200_000_000, "uniform")])     # 1.6 GB kept by the rank: a device copy, not an RCCL message to oneself
def test_cpp_host_runs_the_sharded_schedule_over_rccl(tmp_path, rows, dist_):
    if not os.path.exists(BIN):
        pytest.skip("sharded_host not built")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([BIN, "0", "1", str(tmp_path / "nccl_id"), str(rows), dist_], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["verified"] is True and line["pairs_global"] == rows and line["wire_bytes_per_tuple"] == 12
    assert line["plan"][0] == 2 and line["rowid_mode"] == 3          # rowIDs below 2^32 travel as they are
    assert line["own_segment"] == "device copy" and line["nccl_sends_rank0"] == 0


@pytest.mark.parametrize("rows,dist_,max_msg", [(12_000_000, "uniform", 1 << 20), (12_000_000, "zipf", 3_000_000)])
def test_cpp_host_send_recv_loop_in_pieces_to_self(tmp_path, rows, dist_, max_msg):
    """the grouped ncclSend / ncclRecv loop of sharded_host.cpp with messages of at most max_msg tuples, addressed to oneself"""
    if not os.path.exists(BIN):
        pytest.skip("sharded_host not built")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RHJ_SHARD_VIA_SELF="1", RHJ_SHARD_MAX_MSG=str(max_msg))
    r = subprocess.run([BIN, "0", "1", str(tmp_path / "nccl_id"), str(rows), dist_], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["verified"] is True and line["pairs_global"] == rows
    assert line["own_segment"] == "ncclSend/ncclRecv to self" and line["max_tuples_per_message"] == max_msg
    pieces = -(-rows // max_msg)
    assert line["nccl_sends_rank0"] == 2 * 2 * pieces             # payloads + rowIDs, R and S


def test_cpp_host_peer_mapped_split(tmp_path):
    """RHJ_SHARD_PEER=1: the class split of sharded_host.cpp stores straight into the owner's receive arrays
    (rhj_shard_split_peer): no send buffer, no ncclSend / ncclRecv.  One rank: the owner is the rank itself (the hipIpc leg of
    a multi-rank run cannot execute on a one-GPU box)."""
    if not os.path.exists(BIN):
        pytest.skip("sharded_host not built")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RHJ_SHARD_PEER="1")
    for rows, dist_ in ((12_000_000, "uniform"), (20_000_000, "zipf")):
        r = subprocess.run([BIN, "0", "1", str(tmp_path / "nccl_id"), str(rows), dist_], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["verified"] is True and line["pairs_global"] == rows
        assert line["own_segment"] == "peer-mapped class split" and line["nccl_sends_rank0"] == 0
