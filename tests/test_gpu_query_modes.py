"""GPU suite: whole queries through the `join_gpu` CLI on synthetic relations, host query mode and
device-resident mode, checked against an independent numpy evaluation of the SQL semantics
(SELECT SUM(..) FROM t0,t1,t2 WHERE t0.c1=t1.c0 AND t1.c1=t2.c0 AND t0.c2<X; small.work.sql shape)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOIN = os.path.join(ROOT, "radixhashjoin_amd", "host", "join_gpu")


def write_rel(path, cols):
    with open(path, "wb") as f:
        np.array([len(cols[0]), len(cols)], dtype=np.uint64).tofile(f)
        for c in cols:
            np.ascontiguousarray(c, dtype=np.uint64).tofile(f)


@pytest.mark.parametrize("n,dup", [(300_000, 1), (120_000, 5)])
def test_three_way_join_both_modes(tmp_path, n, dup):
    rng = np.random.default_rng(n)
    # c0 of t1/t2 holds each value `dup` times: joins are many-to-many when dup > 1
    T = []
    for t in range(3):
        T.append([np.arange(n, dtype=np.uint64) // dup, rng.integers(0, n // dup, n, dtype=np.uint64),
                  rng.integers(0, 1000, n, dtype=np.uint64)])
        write_rel(tmp_path / f"t{t}", T[t])
    xs = [100, 450, 999, 0]
    work = "".join(f"0 1 2|0.1=1.0&1.1=2.0&0.2<{x}|0.0 1.2 2.2\n" for x in xs) + "F\n"
    stdin = ("".join(str(tmp_path / f"t{t}") + "\n" for t in range(3)) + "Done\n" + work).encode()
    outs = {}
    for mode in ("host", "device", "batch"):
        r = subprocess.run([JOIN], input=stdin, env=dict(os.environ, RHJ_QUERY_MODE=mode), capture_output=True,
                           check=True, timeout=600)
        outs[mode] = r.stdout.decode().splitlines()
    assert outs["host"] == outs["device"] == outs["batch"]
    # independent evaluation: expand matches with numpy (value v of tX.c0 occupies rows [v*dup, (v+1)*dup))
    for x, line in zip(xs, outs["device"]):
        r0 = np.nonzero(T[0][2] < x)[0]
        if len(r0) == 0:
            assert line == "NULL NULL NULL"
            continue
        r0e = np.repeat(r0, dup)
        r1 = (T[0][1][r0].astype(np.int64)[:, None] * dup + np.arange(dup)[None, :]).ravel()
        r0e2 = np.repeat(r0e, dup)
        r1e = np.repeat(r1, dup)
        r2 = (T[1][1][r1].astype(np.int64)[:, None] * dup + np.arange(dup)[None, :]).ravel()
        exp = [int(T[0][0][r0e2].sum(dtype=np.uint64)), int(T[1][2][r1e].sum(dtype=np.uint64)), int(T[2][2][r2].sum(dtype=np.uint64))]
        assert line == " ".join(str(v) for v in exp)
