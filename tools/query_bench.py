#!/usr/bin/env python3
"""tools/query_bench.py -- whole-query timing of the `join_gpu` CLI on a synthetic star-ish workload that is far
beyond what the reference's query layer can run (its update_intermediate is O(|result| x |intermediate|)).
Three relations of N rows x 3 columns; queries = 1 filter + 2 joins + SUMs.  Compares RHJ_QUERY_MODE=host
(host filters / create_relation / indexed update_intermediate, GPU joins through rhj_join) with
RHJ_QUERY_MODE=device (everything in HBM).  Results of the two modes must be identical."""
import argparse, os, subprocess, sys, tempfile, time
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8_000_000)
ap.add_argument("--queries", type=int, default=8)
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
join = os.path.join(root, "radixhashjoin_amd", "host", "join_gpu")
rng = np.random.default_rng(1)
n = a.rows
with tempfile.TemporaryDirectory() as d:
    for t in range(3):
        cols = [np.arange(n, dtype=np.uint64),                               # c0: key
                rng.integers(0, n, n, dtype=np.uint64),                      # c1: foreign key into another table's c0
                rng.integers(0, 1000, n, dtype=np.uint64)]                   # c2: attribute
        with open(os.path.join(d, f"t{t}"), "wb") as f:
            np.array([n, 3], dtype=np.uint64).tofile(f)
            for c in cols:
                c.tofile(f)
    work = "".join(f"0 1 2|0.1=1.0&1.1=2.0&0.2<{100 + 50 * q}|0.0 1.2 2.2\n" for q in range(a.queries)) + "F\n"
    stdin = ("".join(os.path.join(d, f"t{t}") + "\n" for t in range(3)) + "Done\n" + work).encode()
    outs = {}
    for mode in ("host", "device"):
        t0 = time.perf_counter()
        r = subprocess.run([join], input=stdin, env=dict(os.environ, RHJ_QUERY_MODE=mode), capture_output=True, check=True)
        dt = time.perf_counter() - t0
        outs[mode] = r.stdout
        print(f"{mode:6s}: {dt:7.2f} s for {a.queries} queries over 3 x {n} rows (includes loading the 3 files)", flush=True)
    assert outs["host"] == outs["device"], "modes disagree"
    print("identical results:", outs["host"].decode().splitlines()[:2], "...")
