#!/usr/bin/env python3
"""tests/fuzz_shard.py -- randomized differential test of the multi-GPU stage calls (run by hand on the GPU box): random rank
counts, shard sizes, duplicate structure, plans (fused 2-pass and 17-18 bits), rowID modes and forced join kernels; the union
of the emulated owners' pair sets against the CPU oracle's join of the global relations.  python tests/fuzz_shard.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.pyoracle import Oracle
from radixhashjoin_amd import Engine, Opts
from radixhashjoin_amd.binding import SHARD_GLOBAL16, SHARD_PLAIN, SHARD_TAGGED, shard_plan
from test_gpu_shard_stages import global_relations, sharded_join

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
rng = np.random.default_rng(seed)
o = Oracle()
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    world = int(rng.integers(2, 9))
    n_per = int(rng.choice([1500, 4096, 9000, 20_000, 45_000, 70_000]))
    deep = rng.random() < 0.25
    if deep:
        b1, b2 = [(8, 9), (9, 8), (9, 9)][int(rng.integers(0, 3))]
        mode = SHARD_PLAIN
    else:
        b1, b2 = int(rng.integers(4, 9)), int(rng.integers(1, 9))
        mode = int(rng.choice([SHARD_TAGGED, SHARD_GLOBAL16, SHARD_PLAIN]))
    kernel = 0
    if b1 + b2 >= 16 and rng.random() < 0.7:
        kernel = int(rng.choice([2, 3, 6, 7] + ([4, 5] if mode != SHARD_GLOBAL16 else [])))
    if mode == SHARD_TAGGED and kernel:
        mode = SHARD_GLOBAL16                      # sender tags are resolved by the one-table kernel only
    nlow = int(rng.choice([0, 2, 7])) if b1 + b2 == 16 else 0
    dup = int(rng.choice([1, 1, 3, 50]))
    Rs, Ss = global_relations(rng, world, n_per, nlow, dup, stride=(1 << 30) // 2 if mode == SHARD_PLAIN else (5 << 30))
    sug, plan = shard_plan(n_per, n_per, Opts(2, b1, b2))
    if not sug:
        continue
    e = Engine(0)
    try:
        if kernel:
            e.set_option("join.big_tables", 1)
            e.set_option("join.big_kernel", kernel)
        got = sharded_join(e, Rs, Ss, plan, mode)
    finally:
        e.close()
    exp = o.join(np.concatenate(Rs), np.concatenate(Ss))
    a = got[np.lexsort((got[:, 1], got[:, 0]))] if len(got) else got
    x = np.stack([exp["keyR"], exp["keyS"]], axis=1)
    x = x[np.lexsort((x[:, 1], x[:, 0]))] if len(x) else x
    if len(got) != len(exp) or not np.array_equal(a, x):
        print("MISMATCH", dict(world=world, n_per=n_per, bits=(b1, b2), mode=mode, kernel=kernel, nlow=nlow, dup=dup, got=len(got), exp=len(exp), seed=seed, case=cases), flush=True)
        sys.exit(1)
    cases += 1
print(f"shard fuzz ok: {cases} random sharded joins in {time.time() - t0:.0f} s, seed {seed}")
