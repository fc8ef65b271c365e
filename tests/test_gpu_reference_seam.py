"""GPU suite: the REFERENCE's own Query / intermediate / MainScheduler / CLI code driving the GPU join.

oracle/_ref/join_seam  = every reference file compiled unchanged where it lies; Result::multiRadixHashJoin bound at
                         link time to rhj_join (oracle/ref_gpu_seam.cpp)
oracle/_ref/join_optA  = INTEGRATION.md Option A (Result.cpp / JobScheduler.cpp / half of structs.cpp swapped for the mirror)
oracle/_ref/join_optB  = INTEGRATION.md Option B (only the body of the seam replaced)
All three are built in the build container (`make -C oracle bindings`) and travel as binaries.  Each must print
small/small.result byte for byte.  Their run time is the reference's own update_intermediate (about 3 minutes of
host CPU), so the three run concurrently."""
import collections
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BINARIES = ["join_seam", "join_optA", "join_optB"]


def test_reference_code_drives_gpu_join(tmp_path, small_joins):
    paths = {b: os.path.join(ROOT, "oracle", "_ref", b) for b in BINARIES}
    missing = [b for b, p in paths.items() if not os.path.exists(p)]
    if missing:
        pytest.skip(f"not built (needs the reference checkout at build time): {missing}")
    stdin = open(os.path.join(GOLD, "small", "small.init"), "rb").read() + open(os.path.join(GOLD, "small", "small.work"), "rb").read()
    expected = open(os.path.join(GOLD, "small", "small.result"), "rb").read()
    # the three binaries side by side (most of their minute is the reference's own CPU code above the seam), each driven by
    # communicate() in its own thread: both pipes are drained while the child runs, so a child that writes more than a pipe
    # buffer to stderr cannot block, and the timeout covers the whole exchange
    def run(item):
        b, p = item
        env = dict(os.environ, RHJ_SEAM_LOG=str(tmp_path / f"{b}.log"))
        pr = subprocess.Popen([p], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=GOLD, env=env)
        try:
            out, err = pr.communicate(input=stdin, timeout=900)
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.communicate()
            raise
        return b, pr.returncode, out, err

    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(paths)) as pool:
        for b, rc, out, err in pool.map(run, paths.items()):
            assert rc == 0, (b, err[-2000:])
            assert out == expected, b                             # 50 lines of SUMs / NULLs, byte-identical
    # the seam binary logs every call that went through rhj_join: the same 94 joins the CPU reference makes
    meta, _ = small_joins
    want = collections.Counter((c["nR"], c["nS"], c["count"]) for c in meta)
    got = collections.Counter(tuple(int(x) for x in line.split()) for line in open(tmp_path / "join_seam.log"))
    assert sum(got.values()) == 94 and got == want


def test_reference_code_drives_gpu_join_edge_queries():
    """the same three binaries on the edge-case queries of tests/golden/edge (expected lines from the CPU reference)"""
    paths = {b: os.path.join(ROOT, "oracle", "_ref", b) for b in BINARIES}
    if any(not os.path.exists(p) for p in paths.values()):
        pytest.skip("bindings not built (needs the reference checkout at build time)")
    edge = os.path.join(GOLD, "edge")
    stdin = open(os.path.join(edge, "edge.init"), "rb").read() + open(os.path.join(edge, "edge.work"), "rb").read()
    expected = open(os.path.join(edge, "edge.result"), "rb").read()
    for b, p in paths.items():
        r = subprocess.run([p], input=stdin, cwd=GOLD, capture_output=True, timeout=600)
        assert r.returncode == 0, (b, r.stderr[-2000:])
        assert r.stdout == expected, b
