"""CPU suite: the C-ABI library loads, exports every symbol include/rhj.h declares, the host-side
plan logic behaves, and construction fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

import radixhashjoin_amd as rhj
from radixhashjoin_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rhj.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(rhj_[a-z0-9_]+)\s*\(", src))


def test_library_built_in_tree():
    assert os.path.exists(rhj.lib_path()), "run __graft_entry__.build() / make -C radixhashjoin_amd/csrc"


def test_every_declared_symbol_is_exported_and_bound():
    lib = rhj.load_library()
    declared = header_symbols()
    assert declared == set(binding.SYMBOLS), declared ^ set(binding.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rhj_abi_version() == 3


def test_layouts_match_reference_structs():
    assert rhj.TUPLE.itemsize == 16 and rhj.TUPLE.fields["key"][1] == 0 and rhj.TUPLE.fields["payload"][1] == 8
    assert rhj.PAIR.itemsize == 16 and rhj.PAIR.fields["keyR"][1] == 0 and rhj.PAIR.fields["keyS"][1] == 8


def test_plan_auto():
    P, O = binding.plan, rhj.Opts
    assert P(1, 1561).passes == 0                       # tiny build side: no partitioning (one LDS table)
    assert P(4224, 10**6).passes == 0                   # BJ_CHUNK tuples fit one LDS table
    assert P(4225, 10**6).passes == 1
    assert P(43131, 42987).passes == 0                  # small joins (small.work sizes): unpartitioned, one launch
    assert P(50689, 60000).passes == 1 and P(20000, 131073).passes == 1
    p = P(10**6, 10**6)
    assert (p.passes, p.bits1, p.bits2) == (1, 8, 0)    # BASELINE config 2: 1M x 1M, 8-bit, single pass
    p = P(4 * 10**6, 4 * 10**6)
    assert (p.passes, p.bits1, p.bits2) == (1, 9, 0)    # up to 2.5 8448-tuple chunks per partition: still ONE pass (three launches)
    assert P(10 * 10**6, 10**9).passes == 1 and P(10_200_000, 10_200_000).passes == 2
    p = P(2 * 10**8, 2 * 10**8)
    assert p.passes == 2 and p.bits1 + p.bits2 == 16    # avg build partition 3052 <= 15/16 * 4224
    p = P(10**9, 10**9)
    assert (p.passes, p.bits1, p.bits2) == (2, 8, 8)    # 17-18 bits would re-read the input to count: 16 bits (one fused
                                                        # histogram read) + the compact-table bucket join instead
    p = P(10**9, 10**9, O(2, 8, 8))
    assert (p.passes, p.bits1, p.bits2) == (2, 8, 8)    # BASELINE config 3 as named
    p = P(2 * 10**9, 2 * 10**9)
    assert (p.passes, p.bits1, p.bits2) == (2, 8, 9)    # a 16-bit partition would not fit one compact table any more: 17 bits,
                                                        # the 9-bit pass second (from the narrow intermediate)
    p = P(2_300_000_000, 3 * 10**9)
    assert (p.passes, p.bits1, p.bits2) == (2, 9, 9)    # ... 18 beyond 2.2 * 10^9 (both in the narrow format)
    p = P(8 * 10**9, 8 * 10**9)
    assert (p.passes, p.bits1, p.bits2) == (2, 9, 9)    # never plans a 10-bit pass: the bucket join chunks instead
    assert P(10**9, 5).passes == 0
    for bad in (O(3, 0, 0), O(1, 12, 0), O(2, 8, -1)):
        with pytest.raises(rhj.RhjError):
            P(10, 10, bad)


def test_plan_probe_split_bounds():
    for n in (1, 1000, 43131, 10**6, 10**9):
        ps = binding.plan(n, n).probe_split
        assert 4096 <= ps <= 32768 and ps % 4096 == 0


def test_no_gpu_fails_loudly():
    lib = rhj.load_library()
    if lib.rhj_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rhj.RhjError) as e:
        rhj.Engine(0)
    assert e.value.code == binding.RHJ_E_NODEVICE


def test_product_never_touches_oracle():
    """radixhashjoin_amd/ must not import, link or read anything under oracle/."""
    pkg = os.path.join(ROOT, "radixhashjoin_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("the cpu oracle", ""), os.path.join(dp, f)


def test_shard_plan_modes():
    """rhj_shard_plan: how a multi-GPU receiver gets global rowIDs back, by the kernel that will join its partitions (the same
    choice choose_join_kind makes for a single-GPU join of that size)"""
    from radixhashjoin_amd.binding import SHARD_GLOBAL16, SHARD_PLAIN, SHARD_TAGGED, shard_plan
    m, p = shard_plan(100_000_000, 100_000_000, None)
    assert m == SHARD_TAGGED and (p.bits1, p.bits2) == (8, 7)          # 15 bits: the one-table kernel resolves sender tags
    m, p = shard_plan(200_000_000, 200_000_000, None)
    assert m == SHARD_GLOBAL16 and (p.bits1, p.bits2) == (8, 8)        # 3 K-tuple partitions: compact-table kernel with row guards
    m, p = shard_plan(10**9, 10**9, None)
    assert m == SHARD_GLOBAL16 and (p.bits1, p.bits2) == (8, 8)
    m, p = shard_plan(1_500_000_000, 1_500_000_000, None)
    assert m == SHARD_PLAIN and (p.bits1, p.bits2) == (8, 9)           # 17 bits: only rowIDs that need no restoring
    m, _ = shard_plan(1_000_000, 1_000_000, None)
    assert m == 0                                                      # one-pass plans: not the narrow sharded path
