"""CPU suite: host logic of the C++ mirror that needs no GPU -- query text parser, Result page
bookkeeping, update_intermediate (3 cases) against a brute-force restatement of the reference's
semantics (radixhashjoin_amd/host/query_unit.cpp), and the build products exist."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "radixhashjoin_amd", "host")


def test_query_layer_unit_checks():
    exe = os.path.join(HOST, "query_unit")
    assert os.path.exists(exe), "build with __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "all checks passed" in r.stdout


def test_scheduler_mirror_survives_start_stop_stress():
    """radixhashjoin_amd/host/sched_stress.cpp: 20000 schedulers started and stopped at once (idle workers: the window in
    which the reference's lock-free `done = true` in JobScheduler::stop, JobScheduler.cpp:140-146, loses its wake-up -- see
    oracle/ref_sched_race.cpp), every eighth one with jobs, a barrier and jobs still queued at stop()"""
    exe = os.path.join(HOST, "sched_stress")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-s", "-C", HOST, "sched_stress"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe, "20000", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok 20000 cycles")


def test_host_binaries_built():
    for name in ("librhj_compat.a", "host_driver", "join_gpu"):
        assert os.path.exists(os.path.join(HOST, name)), name


def test_join_cli_fails_loudly_without_gpu():
    """no CPU fallback: on a box without a GPU the CLI must exit non-zero with the C-ABI's error"""
    import radixhashjoin_amd as rhj
    if rhj.load_library().rhj_device_count() > 0:
        import pytest
        pytest.skip("a GPU is present")
    gold = os.path.join(ROOT, "tests", "golden")
    stdin = open(os.path.join(gold, "small", "small.init"), "rb").read() + b"3 0|0.2=1.0|1.2\nF\n"
    r = subprocess.run([os.path.join(HOST, "join_gpu")], input=stdin, cwd=gold, capture_output=True, timeout=120)
    assert r.returncode != 0
    assert b"rhj_init failed" in r.stderr


def test_balanced_owner_cuts():
    """radixhashjoin_amd.sharded.balanced_cuts: contiguous class ranges of near-equal weight, computed by every
    rank from the same all-gathered histogram"""
    from radixhashjoin_amd.sharded import balanced_cuts
    assert balanced_cuts([1] * 16, 4) == [0, 4, 8, 12, 16]
    assert balanced_cuts([10, 1, 1, 1, 1, 1, 1, 1], 2) == [0, 1, 8]              # the hot class alone is more than half
    c = balanced_cuts([0, 0, 0, 5, 0, 0], 3)
    assert c[0] == 0 and c[-1] == 6 and c == sorted(c)                           # ranges may be empty, never reversed
    import random
    rng = random.Random(3)
    for world in (2, 3, 4, 8):
        w = [rng.randint(0, 1000) for _ in range(256)]
        w[17] = 20_000                                                           # one hot class
        cuts = balanced_cuts(w, world)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == 256 and cuts == sorted(cuts)
        loads = [sum(w[cuts[r]:cuts[r + 1]]) for r in range(world)]
        assert sum(loads) == sum(w)
        # no rank exceeds the mean by more than the largest single class (a class cannot be split)
        assert max(loads) <= sum(w) / world + max(w)
