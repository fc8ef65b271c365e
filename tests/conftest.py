import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test, skipped unless RHJ_SLOW=1")


def pytest_sessionstart(session):
    """Built artefacts are git-ignored: if this checkout has not been built yet, build it (hipcc cross-compiles
    gfx950 without a GPU; the oracle needs only gcc)."""
    need = [os.path.join(ROOT, "radixhashjoin_amd", "librhj_hip.so"), os.path.join(ROOT, "oracle", "liborc.so"),
            os.path.join(ROOT, "radixhashjoin_amd", "host", "join_gpu"),
            os.path.join(ROOT, "radixhashjoin_amd", "host", "host_driver"),
            os.path.join(ROOT, "radixhashjoin_amd", "host", "query_unit"),
            os.path.join(ROOT, "radixhashjoin_amd", "host", "sharded_host")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    if os.environ.get("RHJ_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set RHJ_SLOW=1 to run")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/rhj_oracle.c) -- the checker, never the thing under test on GPU."""
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference compiled into oracle/_ref (present in the build container and, as a
    prebuilt binary, on the GPU box); tests that need it skip when it is absent."""
    from oracle.pyoracle import Reference, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref/libref_rhj.so not built")
    return Reference()


@pytest.fixture(scope="session")
def synthetic_golden():
    with open(os.path.join(GOLDEN, "synthetic.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def small_joins():
    with open(os.path.join(GOLDEN, "small_joins.json")) as f:
        meta = json.load(f)["calls"]
    return meta, np.load(os.path.join(GOLDEN, "small_joins.npz"))


@pytest.fixture(scope="session")
def tiny_vectors():
    return np.load(os.path.join(GOLDEN, "tiny_vectors.npz"))


@pytest.fixture(scope="session")
def engine():
    """One rhj_ctx on cuda:0 through the C-ABI.  Fails loudly if the HIP library is missing."""
    from radixhashjoin_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def make_inputs(o, spec):
    """inputs of a tests/golden/synthetic.json case (generators of SURVEY.md App. A)"""
    if spec["S"] == "const":
        return o.gen_const(spec["nR"], spec["value"]), o.gen_const(spec["nS"], spec["value"])
    R = o.gen_R(spec["nR"], spec["D"])
    if spec["S"] == "chain":
        return R, o.gen_S_chain(spec["nS"], spec["D"])
    if spec["S"] == "disjoint":
        return R, o.gen_S_disjoint(spec["nS"], spec["D"])
    raise ValueError(spec["S"])


def small_call_arrays(npz, i):
    from oracle.pyoracle import PAIR, TUPLE
    def tup(tag, a, b, dt):
        x = np.empty(len(npz[f"call{i}__{tag}_{a}"]), dtype=dt)
        x[a] = npz[f"call{i}__{tag}_{a}"]
        x[b] = npz[f"call{i}__{tag}_{b}"]
        return x
    return tup("R", "key", "payload", TUPLE), tup("S", "key", "payload", TUPLE), tup("P", "keyR", "keyS", PAIR)
