"""ctypes loaders for the CPU oracle (oracle/liborc.so) and, when built, the real reference
(oracle/_ref/libref_rhj.so).  TEST INFRASTRUCTURE: import only from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke().  Never imported by radixhashjoin_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TUPLE = np.dtype([("key", "<u8"), ("payload", "<u8")])     # reference structs.h:33-36
PAIR = np.dtype([("keyR", "<u8"), ("keyS", "<u8")])        # reference Result.h:9-12

_vp, _sz, _u64 = C.c_void_p, C.c_size_t, C.c_uint64


def build(force=False):
    """Compile liborc.so (and oracle/_ref when /root/reference exists). Building != using."""
    so = os.path.join(HERE, "liborc.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "rhj_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "liborc.so"])
    if os.path.isdir(os.environ.get("RHJ_REFERENCE", "/root/reference")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _ptr(a):
    return a.ctypes.data_as(_vp)


def as_tuples(keys, payloads):
    t = np.empty(len(keys), dtype=TUPLE)
    t["key"] = keys
    t["payload"] = payloads
    return t


class Oracle:
    """The CPU restatement (rhj_oracle.c)."""

    def __init__(self):
        build()
        # RHJ_ORACLE_LIB: another build of the same source (`make -C oracle asan` runs the oracle's tests over liborc_asan.so)
        self.lib = L = C.CDLL(os.environ.get("RHJ_ORACLE_LIB") or os.path.join(HERE, "liborc.so"))
        L.orc_next_prime.restype = _sz
        L.orc_next_prime.argtypes = [_sz]
        L.orc_mix.restype = _u64
        L.orc_mix.argtypes = [_u64]
        L.orc_hash_relation.argtypes = [_vp, _sz, _sz, C.c_int, _vp, _vp]
        L.orc_single_partition.argtypes = [_vp, _sz, _sz, _vp, _vp]
        L.orc_join_flat.restype = _sz
        L.orc_join_flat.argtypes = [_vp, _sz, _vp, _sz, C.POINTER(_vp)]
        L.orc_join_count_checksum.restype = _sz
        L.orc_join_count_checksum.argtypes = [_vp, _sz, _vp, _sz, C.POINTER(_u64)]
        L.orc_pairs_checksum.restype = _u64
        L.orc_pairs_checksum.argtypes = [_vp, _sz]
        for f in ("orc_gen_R", "orc_gen_S_chain", "orc_gen_S_disjoint", "orc_gen_const"):
            getattr(L, f).argtypes = [_vp, _sz, _u64]
        L.orc_gen_S_counter.argtypes = [_vp, _sz, _u64, _u64]
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [_vp]

    def next_prime(self, x):
        return self.lib.orc_next_prime(x)

    def mix(self, z):
        return self.lib.orc_mix(z)

    def hash_relation(self, rel, nbins=256, nranges=8):
        rel = np.ascontiguousarray(rel, dtype=TUPLE)
        out = np.empty(len(rel), dtype=TUPLE)
        hist = np.zeros(nbins, dtype=np.uint64)
        self.lib.orc_hash_relation(_ptr(rel), len(rel), nbins, nranges, _ptr(out), _ptr(hist))
        return out, hist

    def single_partition(self, rel, nbins=256):
        rel = np.ascontiguousarray(rel, dtype=TUPLE)
        out = np.empty(len(rel), dtype=TUPLE)
        hist = np.zeros(nbins, dtype=np.uint64)
        self.lib.orc_single_partition(_ptr(rel), len(rel), nbins, _ptr(out), _ptr(hist))
        return out, hist

    def join(self, R, S):
        """-> pairs array (PAIR dtype) in the reference's page order."""
        R = np.ascontiguousarray(R, dtype=TUPLE)
        S = np.ascontiguousarray(S, dtype=TUPLE)
        p = _vp()
        n = self.lib.orc_join_flat(_ptr(R), len(R), _ptr(S), len(S), C.byref(p))
        out = np.empty(n, dtype=PAIR)
        if n:
            C.memmove(out.ctypes.data, p, n * PAIR.itemsize)
        self.libc.free(p)
        return out

    def join_count_checksum(self, R, S):
        R = np.ascontiguousarray(R, dtype=TUPLE)
        S = np.ascontiguousarray(S, dtype=TUPLE)
        c = _u64()
        n = self.lib.orc_join_count_checksum(_ptr(R), len(R), _ptr(S), len(S), C.byref(c))
        return n, c.value

    def pairs_checksum(self, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=PAIR)
        return self.lib.orc_pairs_checksum(_ptr(pairs), len(pairs))

    # generators of SURVEY.md App. A
    def gen_R(self, n, D=None):
        t = np.empty(n, dtype=TUPLE)
        self.lib.orc_gen_R(_ptr(t), n, D if D is not None else max(n, 1))
        return t

    def gen_S_chain(self, n, D):
        t = np.empty(n, dtype=TUPLE)
        self.lib.orc_gen_S_chain(_ptr(t), n, D)
        return t

    def gen_S_disjoint(self, n, D):
        t = np.empty(n, dtype=TUPLE)
        self.lib.orc_gen_S_disjoint(_ptr(t), n, D)
        return t

    def gen_const(self, n, value):
        t = np.empty(n, dtype=TUPLE)
        self.lib.orc_gen_const(_ptr(t), n, value)
        return t

    def gen_S_counter(self, n, D, seed):
        t = np.empty(n, dtype=TUPLE)
        self.lib.orc_gen_S_counter(_ptr(t), n, D, seed)
        return t


def _ref_lib(threads):
    return os.path.join(HERE, "_ref", "libref_rhj.so" if threads == 8 else f"libref_rhj_{threads}t.so")


def ref_available(threads=8):
    return os.path.exists(_ref_lib(threads))


class Reference:
    """The real reference, compiled by oracle/Makefile into oracle/_ref/ (binary only).  threads = 8 is the reference as
    shipped (NUM_OF_THREADS, JobScheduler.h:11); threads = 1 the build with that one macro patched in a /tmp copy of the
    header (the CPU-baseline protocol times both)."""

    def __init__(self, threads=8):
        self.lib = L = C.CDLL(_ref_lib(threads))
        L.ref_num_threads.restype = C.c_int
        L.ref_next_prime.restype = _sz
        L.ref_next_prime.argtypes = [_sz]
        L.ref_join.restype = _sz
        L.ref_join.argtypes = [_vp, _sz, _vp, _sz, C.POINTER(_vp), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.ref_hash_relation.argtypes = [_vp, _sz, _sz, _vp, _vp]
        L.ref_single_partition.argtypes = [_vp, _sz, _sz, _vp, _vp]
        L.ref_free.argtypes = [_vp]
        self.num_threads = L.ref_num_threads()

    def next_prime(self, x):
        return self.lib.ref_next_prime(x)

    def join(self, R, S, want_pairs=True):
        """-> (pairs or None, count, head_is_null, seconds of multiRadixHashJoin)"""
        R = np.ascontiguousarray(R, dtype=TUPLE)
        S = np.ascontiguousarray(S, dtype=TUPLE)
        p, hn, sec = _vp(), C.c_int(), C.c_double()
        n = self.lib.ref_join(_ptr(R), len(R), _ptr(S), len(S),
                              C.byref(p) if want_pairs else None, C.byref(hn), C.byref(sec))
        out = None
        if want_pairs:
            out = np.empty(n, dtype=PAIR)
            if n:
                C.memmove(out.ctypes.data, p, n * PAIR.itemsize)
            self.lib.ref_free(p)
        return out, n, bool(hn.value), sec.value

    def hash_relation(self, rel, nbins=256):
        rel = np.ascontiguousarray(rel, dtype=TUPLE)
        out = np.empty(len(rel), dtype=TUPLE)
        hist = np.zeros(nbins, dtype=np.uint64)
        self.lib.ref_hash_relation(_ptr(rel), len(rel), nbins, _ptr(out), _ptr(hist))
        return out, hist

    def single_partition(self, rel, nbins=256):
        rel = np.ascontiguousarray(rel, dtype=TUPLE)
        out = np.empty(len(rel), dtype=TUPLE)
        hist = np.zeros(nbins, dtype=np.uint64)
        self.lib.ref_single_partition(_ptr(rel), len(rel), nbins, _ptr(out), _ptr(hist))
        return out, hist


def sorted_pairs(p):
    """canonical order for order-insensitive comparison"""
    p = np.ascontiguousarray(p, dtype=PAIR)
    return p[np.lexsort((p["keyS"], p["keyR"]))]
