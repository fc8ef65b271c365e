"""GPU suite: BASELINE config 1 -- the reference's own golden workload small/small.init + small.work
through the GPU engine (radixhashjoin_amd/host/join_gpu = the reference's CLI protocol on top of
rhj_compat / rhj_query), byte-identical to the reference's small/small.result; and the 94 hot-path
calls it makes are the 94 calls the reference makes (tests/golden/small_joins.json, link-time tap)."""
import collections
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
JOIN = os.path.join(ROOT, "radixhashjoin_amd", "host", "join_gpu")


def run_small(tmp_path, mode="host"):
    assert os.path.exists(JOIN), "build with __graft_entry__.build()"
    log = tmp_path / "joins.log"
    env = dict(os.environ, RHJ_JOIN_LOG=str(log), RHJ_QUERY_MODE=mode)
    stdin = open(os.path.join(GOLD, "small", "small.init"), "rb").read() + open(os.path.join(GOLD, "small", "small.work"), "rb").read()
    out = subprocess.run([JOIN], input=stdin, cwd=GOLD, env=env, capture_output=True, timeout=600, check=True).stdout
    return out, log


def test_small_work_byte_identical(tmp_path, small_joins):
    out, log = run_small(tmp_path)
    expected = open(os.path.join(GOLD, "small", "small.result"), "rb").read()
    assert out == expected                                        # 50 lines of SUMs / NULLs
    meta, _ = small_joins
    want = collections.Counter((c["nR"], c["nS"], c["count"]) for c in meta)
    got = collections.Counter(tuple(int(x) for x in line.split()) for line in open(log))
    assert sum(got.values()) == 94
    assert got == want                                            # the same 94 joins, same sizes, same match counts


def test_small_work_device_resident_queries(tmp_path):
    """the same workload with the whole query device-resident (RHJ_QUERY_MODE=device: columns in HBM, device
    filters, position-carrying join inputs, gathered intermediates, device SUMs): same 50 lines"""
    out, _ = run_small(tmp_path, mode="device")
    assert out == open(os.path.join(GOLD, "small", "small.result"), "rb").read()


@pytest.mark.parametrize("mode", ["host", "device"])
def test_edge_queries_match_the_reference(mode):
    """corners of Query::run_joins that small.work never reaches: a projected alias that is never joined (sums to 0,
    Query.cpp:198-200 over an empty intermediate), two disconnected joins (the second drops the older columns,
    intermediate.cpp:147-162), the a-b / c-d / b-c order, a predicate between two aliases already joined.  Expected
    lines were printed by the REAL reference (tests/golden/make_edge.py); same-alias predicates are parity-unpinned (the
    reference segfaults on them)."""
    edge = os.path.join(GOLD, "edge")
    stdin = open(os.path.join(edge, "edge.init"), "rb").read() + open(os.path.join(edge, "edge.work"), "rb").read()
    env = dict(os.environ, RHJ_QUERY_MODE=mode)
    out = subprocess.run([JOIN], input=stdin, cwd=GOLD, env=env, capture_output=True, timeout=600, check=True).stdout
    assert out == open(os.path.join(edge, "edge.result"), "rb").read()
