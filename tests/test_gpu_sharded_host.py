"""GPU suite: the multi-GPU join from a plain C++ host (radixhashjoin_amd/host/sharded_host.cpp): RCCL collectives
(ncclAllGather, grouped ncclSend / ncclRecv, ncclAllReduce) + the rhj_shard_* stage calls of include/rhj.h, no Python and no
torch.  A gpurun box has one GPU, so this is the one-rank run of the program an 8-GPU node would start eight times: every
RCCL and C-ABI call of the schedule executes (the exchange is addressed to oneself); the program verifies its own pair set
(count + checksum against the closed form, all-reduced)."""
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
