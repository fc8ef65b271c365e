"""GPU suite: the C++ host mirror of the reference surface (radixhashjoin_amd/host/rhj_compat.h)
driven the way the reference's own code drives it (host_driver.cpp), checked against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle.pyoracle import PAIR, TUPLE, sorted_pairs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "radixhashjoin_amd", "host", "host_driver")


def run(args):
    assert os.path.exists(DRIVER), "build with __graft_entry__.build() / make -C radixhashjoin_amd/host"
    return subprocess.run([DRIVER] + args, check=True, capture_output=True, text=True, timeout=300).stdout


CASES = [(50_000, 80_000, 20_000), (1, 1561, 1), (300_000, 200_000, 300_000), (7, 7, 7), (10_000, 10_000, 100)]


@pytest.mark.parametrize("mode", ["direct", "staged"])
@pytest.mark.parametrize("nR,nS,D", CASES)
def test_result_surface(oracle, tmp_path, mode, nR, nS, D):
    R, S = oracle.gen_R(nR, D), oracle.gen_S_chain(nS, D)
    R.tofile(tmp_path / "R.bin"); S.tofile(tmp_path / "S.bin")
    out = run([mode, str(tmp_path / "R.bin"), str(tmp_path / "S.bin"), str(tmp_path / "out.bin")])
    got = np.fromfile(tmp_path / "out.bin", dtype=PAIR)
    exp = oracle.join(R, S)
    assert np.array_equal(sorted_pairs(got), sorted_pairs(exp))
    m = re.search(r"matches=(\d+) head=(\w+) capacity=(\d+) size=(\d+)", out)
    assert int(m.group(1)) == len(exp) and m.group(2) == "set"
    if mode == "direct":           # one device-filled page: capacity == size == matches
        assert int(m.group(3)) == int(m.group(4)) == len(exp)


def test_empty_result_keeps_head_null(oracle, tmp_path):
    R, S = oracle.gen_R(1000), oracle.gen_S_disjoint(1000, 1000)
    R.tofile(tmp_path / "R.bin"); S.tofile(tmp_path / "S.bin")
    out = run(["direct", str(tmp_path / "R.bin"), str(tmp_path / "S.bin"), str(tmp_path / "out.bin")])
    assert "head=null" in out and os.path.getsize(tmp_path / "out.bin") == 0
    assert "capacity=8191 size=8191" in out            # untouched default-constructed Result (Result.cpp:10-14)


def test_histogram_and_partition_jobs(oracle, tmp_path):
    n = 100_003
    R = oracle.gen_R(n, n // 3)
    R.tofile(tmp_path / "R.bin")
    run(["jobs", str(tmp_path / "R.bin"), str(tmp_path / "jobs.bin")])
    raw = np.fromfile(tmp_path / "jobs.bin", dtype=np.uint64)
    q, r = divmod(n, 8)
    start = 0
    pos = 0
    for i in range(8):
        cnt = q + (1 if 1 <= i <= r else 0)                       # structs.cpp:146-161
        assert raw[pos] == cnt; pos += 1
        hist = raw[pos:pos + 256]; pos += 256
        sums = raw[pos:pos + 256]; pos += 256
        idx = raw[pos:pos + cnt]; pos += cnt
        d = (R["payload"][start:start + cnt] & np.uint64(255)).astype(np.int64)
        exp_hist = np.bincount(d, minlength=256)
        assert np.array_equal(hist.astype(np.int64), exp_hist)                       # HistogramJob::run
        assert np.array_equal(sums.astype(np.int64), np.concatenate([[0], np.cumsum(exp_hist)[:-1]]))
        # PartitionJob: global row indices grouped by bucket (order inside a bucket unspecified)
        got_d = (R["payload"][idx.astype(np.int64)] & np.uint64(255)).astype(np.int64)
        assert np.all(np.diff(got_d) >= 0)
        assert np.array_equal(np.sort(idx), np.arange(start, start + cnt, dtype=np.uint64))
        start += cnt
    assert pos == len(raw)
