#!/usr/bin/env python3
"""tests/golden/make_edge.py -- regenerates tests/golden/edge/: four small synthetic relations in the reference's binary
format (structs.cpp:28-39: [num_tuples, num_columns] then column-major uint64), six queries that exercise the corners of
Query::run_joins the golden workload small.work never reaches (a projected alias that is never joined, two disconnected
joins, the a-b / c-d / b-c order, a predicate between two aliases already in the intermediate), and edge.result = what the
REAL reference (oracle/_ref/join_ref, compiled from its own sources by oracle/Makefile) prints for them.
Same-alias predicates (parse_table, intermediate.cpp:11-44) are NOT covered: the reference segfaults on them here
(SURVEY §2 #7 documents the undefined behaviour), so that corner is parity-unpinned."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "edge")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(7)
for name, n in (("e0", 120), ("e1", 200), ("e2", 90), ("e3", 150)):
    a = np.empty((3, n), dtype=np.uint64)
    for c in range(3):
        a[c] = rng.integers(0, 20, n)
    with open(os.path.join(OUT, name), "wb") as f:
        f.write(np.array([n, 3], dtype=np.uint64).tobytes())
        f.write(a.tobytes())
open(os.path.join(OUT, "edge.init"), "w").write("./edge/e0\n./edge/e1\n./edge/e2\n./edge/e3\nDone\n")
queries = ["0 1 2|0.0=1.0|2.1 0.1", "0 1 2 3|0.0=1.0&2.0=3.0|0.1 2.1 3.2", "0 1 2 3|0.0=1.0&2.0=3.0&1.1=2.1|0.1 3.2",
           "0 1|0.0=1.0&0.1>5|1.2 0.2", "0 1 2|0.0=1.0&1.1=2.1&0.2<10|0.1 1.2 2.0", "0 1|0.0=1.0&0.1=1.1|0.2 1.2"]
open(os.path.join(OUT, "edge.work"), "w").write("\n".join(queries) + "\nF\n")
ref = os.path.join(HERE, "..", "..", "oracle", "_ref", "join_ref")
stdin = open(os.path.join(OUT, "edge.init"), "rb").read() + open(os.path.join(OUT, "edge.work"), "rb").read()
out = subprocess.run([ref], input=stdin, cwd=HERE, capture_output=True, check=True).stdout
open(os.path.join(OUT, "edge.result"), "wb").write(out)
print(out.decode())
