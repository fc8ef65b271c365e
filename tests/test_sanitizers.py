"""CPU suite: the sanitizer builds of the host side (SURVEY §5 "race detection / sanitizers").  GPU sanitizers are not available
on this pool, and the device code has no host-visible threads anyway; what runs real concurrency on the host is
  * the scheduler mirror's worker pool (radixhashjoin_amd/host/rhj_compat.cpp)          -> `make -C radixhashjoin_amd/host -f Makefile.sanitize tsan`
  * rhj_join's six stager workers, its downloader and its page pre-faulters, under several query threads with a context
    each (radixhashjoin_amd/csrc/rhj_api.hip, compiled as C++ over a host-only stand-in for the HIP runtime)
                                                                                         -> `make -C radixhashjoin_amd/csrc -f Makefile.sanitize tsan`
  * and the checker itself: the oracle under ASan + UBSan, over its own test file       -> `make -C oracle -f Makefile.sanitize asan`
A target fails on any sanitizer report (TSan exit code 66, ASan abort) or on a wrong result page."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_target(directory, target, timeout):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), "-f", "Makefile.sanitize", target], capture_output=True, text=True,
                       timeout=timeout)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ThreadSanitizer" not in tail and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    return r.stdout


def test_scheduler_mirror_under_tsan():
    assert "ok 3000 cycles" in run_target("radixhashjoin_amd/host", "tsan", 600)


def test_rhj_join_host_threads_under_tsan():
    assert "every page as expected" in run_target("radixhashjoin_amd/csrc", "tsan", 900)


def test_oracle_under_asan_ubsan():
    assert " passed" in run_target("oracle", "asan", 900)
