"""The documented reference-side bindings really build (INTEGRATION.md Option A and Option B).

CPU test: tools/bind_reference.sh applies each option to a /tmp copy of the reference checkout exactly as
INTEGRATION.md describes and compiles AND links the reference's `join` binary against librhj_hip.so
(+ librhj_compat.a for Option A).  No GPU is needed to link.  Skipped where /root/reference is absent
(the GPU box): there the prebuilt oracle/_ref/join_{seam,optA,optB} are run by test_gpu_reference_seam.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RHJ_REFERENCE", "/root/reference")
TOOL = os.path.join(ROOT, "tools", "bind_reference.sh")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout absent")


def _undefined_symbols(binary):
    out = subprocess.run(["nm", "-D", "--undefined-only", binary], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}


def _defined_symbols(binary):
    out = subprocess.run(["nm", "-C", "--defined-only", binary], capture_output=True, text=True, check=True).stdout
    return out


@pytest.mark.parametrize("option", ["A", "B"])
def test_binding_compiles_and_links(tmp_path, option):
    out = tmp_path / f"opt{option}"
    r = subprocess.run([TOOL, option, REF, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    binary = out / "join"
    assert binary.exists()
    und = _undefined_symbols(str(binary))
    assert "rhj_join" in und and "rhj_init" in und            # the seam is bound to the C-ABI, not to a CPU body
    defined = _defined_symbols(str(binary))
    # everything above the seam is the reference's own code
    for sym in ("Query::run_joins", "update_intermediate", "mainThreadWork", "MainScheduler::init"):
        assert sym in defined, sym
    if option == "A":
        # the reference's MainScheduler.cpp:6-18 compiled against the mirror: these are the members it needs
        assert "JobScheduler::threadWork(void*)" in defined
        assert "JobScheduler::init(unsigned long, void* (*)(void*))" in defined
        assert "HistogramJob::run" in defined and "rhj_histogram" in und     # job bodies forward to the device
    else:
        # Option B keeps every reference file; only the seam's body changed
        assert "JoinJob::run" in defined and "Result::join_buckets" in defined


def test_seam_target_builds():
    """oracle/Makefile `bindings`: reference objects unchanged + a link-time seam on the mangled symbol"""
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "bindings"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    seam = os.path.join(ROOT, "oracle", "_ref", "join_seam")
    assert os.path.exists(seam)
    assert "rhj_join" in _undefined_symbols(seam)
    d = _defined_symbols(seam)
    assert "Result::refMultiRadixHashJoin" in d               # the CPU body is still there under another name...
    assert "Query::run_joins" in d


def test_bound_binaries_carry_the_fixed_stop(tmp_path):
    """join_seam binds JobScheduler::stop next to the seam (the reference's racy body stays under another name, uncalled);
    Option B's copy of JobScheduler.cpp gets the two-line source edit and nothing else; and the start / stop loop over the
    fixed stop() gets through 2 x 10^5 cycles (the reference's own stalls within 10^5 here: `make -C oracle _ref/sched_race`)"""
    seam = os.path.join(ROOT, "oracle", "_ref", "join_seam")
    d = _defined_symbols(seam)
    assert "JobScheduler::stop()" in d and "JobScheduler::ref_stop_racy()" in d
    out = tmp_path / "optB"
    r = subprocess.run([TOOL, "B", REF, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    diff = subprocess.run(["diff", os.path.join(REF, "JobScheduler.cpp"), str(out / "JobScheduler.cpp")], capture_output=True, text=True).stdout
    changed = [ln for ln in diff.splitlines() if ln[:1] in "<>"]
    assert changed == [">     pthread_mutex_lock(&queueLock);", ">     pthread_mutex_unlock(&queueLock);"], diff
    loop = os.path.join(ROOT, "oracle", "_ref", "sched_race_fixed")
    r = subprocess.run([loop, "200000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("completed")


def test_cpp_sharded_host_links_rccl_and_the_shard_stage_calls():
    """radixhashjoin_amd/host/sharded_host.cpp: the multi-GPU schedule from a C++ host -- RCCL for the collectives, the
    C-ABI for the compute (no torch, no Python)"""
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "radixhashjoin_amd", "host"), "sharded_host"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    binary = os.path.join(ROOT, "radixhashjoin_amd", "host", "sharded_host")
    und = _undefined_symbols(binary)
    for sym in ("rhj_shard_stats", "rhj_shard_split", "rhj_shard_partition", "rhj_shard_join", "rhj_shard_plan",
                "ncclAllGather", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclAllReduce", "ncclCommInitRank"):
        assert sym in und, sym
