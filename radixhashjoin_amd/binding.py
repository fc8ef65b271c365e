"""ctypes binding of include/rhj.h (librhj_hip.so).  No compute happens in Python."""
import ctypes as C
import os

import numpy as np

TUPLE = np.dtype([("key", "<u8"), ("payload", "<u8")])   # rhj_tuple == reference `tuple` (structs.h:33-36)
PAIR = np.dtype([("keyR", "<u8"), ("keyS", "<u8")])      # rhj_pair  == reference `key_tuple` (Result.h:9-12)

RHJ_OK, RHJ_E_INVALID, RHJ_E_NODEVICE, RHJ_E_HIP, RHJ_E_NOMEM, RHJ_E_OVERFLOW = 0, -1, -2, -3, -4, -5
KERNEL_KINDS = ("hist", "scan", "scatter", "tasks", "join", "aux")

_vp, _u64, _i32 = C.c_void_p, C.c_uint64, C.c_int32


class Opts(C.Structure):
    """rhj_opts: passes=-1 auto; bits per pass; probe_split = max probe tuples per join task."""
    _fields_ = [("passes", _i32), ("bits1", _i32), ("bits2", _i32), ("probe_split", _i32)]

    def __init__(self, passes=-1, bits1=0, bits2=0, probe_split=0):
        super().__init__(passes, bits1, bits2, probe_split)

    def __repr__(self):
        return f"Opts(passes={self.passes}, bits1={self.bits1}, bits2={self.bits2}, probe_split={self.probe_split})"


class JoinDesc(C.Structure):
    """rhj_join_desc (include/rhj.h): one join of an rhj_join_batch call"""
    _fields_ = [("R", C.c_void_p), ("nR", C.c_uint64), ("S", C.c_void_p), ("nS", C.c_uint64)]


class Timings(C.Structure):
    _fields_ = [("ms", C.c_double * 6), ("launches", C.c_uint32 * 6), ("total_ms", C.c_double),
                ("passes", _i32), ("bits1", _i32), ("bits2", _i32), ("ntasks", _u64)]

    def as_dict(self):
        d = {k: {"ms": self.ms[i], "launches": self.launches[i]} for i, k in enumerate(KERNEL_KINDS)}
        d.update(total_ms=self.total_ms, passes=self.passes, bits1=self.bits1, bits2=self.bits2, ntasks=self.ntasks)
        return d


class RhjError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rhj error {code}: {msg}")
        self.code = code


def lib_path():
    """the in-tree library; RHJ_LIB_PATH names another build of it (development A/B runs: tools/ab/)"""
    return os.environ.get("RHJ_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librhj_hip.so")


_LIB = None

# every symbol include/rhj.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "rhj_abi_version": (C.c_int, []),
    "rhj_device_count": (C.c_int, []),
    "rhj_init": (C.c_int, [C.c_int, _P(_vp)]),
    "rhj_destroy": (None, [_vp]),
    "rhj_last_error": (C.c_char_p, [_vp]),
    "rhj_set_stream": (C.c_int, [_vp, _vp]),
    "rhj_set_profiling": (C.c_int, [_vp, C.c_int]),
    "rhj_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "rhj_get_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "rhj_get_timings": (C.c_int, [_vp, _P(Timings)]),
    "rhj_get_launch_timings": (C.c_int, [_vp, _P(_i32), _P(C.c_double), C.c_uint32, _P(C.c_uint32)]),
    "rhj_sync": (C.c_int, [_vp]),
    "rhj_reserve": (C.c_int, [_vp, _u64, _u64, _P(Opts)]),
    "rhj_release_workspace": (C.c_int, [_vp]),
    "rhj_default_opts": (None, [_P(Opts)]),
    "rhj_plan": (C.c_int, [_u64, _u64, _P(Opts), _P(Opts)]),
    "rhj_join": (C.c_int, [_vp, _vp, _u64, _vp, _u64, _P(Opts), _P(_vp), _P(_u64)]),
    "rhj_join_batch": (C.c_int, [_vp, C.c_uint32, _P(JoinDesc), _P(_vp), _P(_u64)]),
    "rhj_join_dev": (C.c_int, [_vp, _vp, _u64, _vp, _u64, _P(Opts), _vp, _u64, _P(_u64)]),
    "rhj_histogram": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, _vp]),
    "rhj_prefix": (C.c_int, [_vp, _vp, _u64, _vp]),
    "rhj_partition": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, _vp, _vp]),
    "rhj_partition_at": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, _vp, _vp]),
    "rhj_owner_histogram": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, _vp]),
    "rhj_owner_split": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, _vp, _vp]),
    "rhj_mix64": (_u64, [_u64]),
    "rhj_unmix64": (_u64, [_u64]),
    "rhj_bucket_join": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u64, C.c_int, C.c_int, _vp, _u64, _P(_u64)]),
    "rhj_narrow_key_offset": (_u64, [_u64]),
    "rhj_narrow_bytes": (_u64, [_u64]),
    "rhj_shard_plan": (C.c_int, [_u64, _u64, _P(Opts), _P(Opts)]),
    "rhj_shard_stats": (C.c_int, [_vp, C.c_int, _vp, _u64, C.c_int, C.c_int, _vp, _P(_u64), _P(_u64)]),
    "rhj_shard_split": (C.c_int, [_vp, C.c_int, _vp, _u64, C.c_int, C.c_int, _u64, _vp, _vp]),
    "rhj_shard_split_peer": (C.c_int, [_vp, C.c_int, _vp, _u64, C.c_int, C.c_int, _u64, _vp, _vp, _vp, _vp, C.c_int]),
    "rhj_ipc_export": (C.c_int, [_vp, _vp, _vp]),
    "rhj_ipc_open": (C.c_int, [_vp, _vp, _P(_vp)]),
    "rhj_ipc_close": (C.c_int, [_vp, _vp]),
    "rhj_shard_partition": (C.c_int, [_vp, C.c_int, _vp, _vp, _u64, C.c_int, _vp, _vp, _P(Opts), C.c_int]),
    "rhj_shard_join": (C.c_int, [_vp, _vp, _u64, _P(_u64)]),
    "rhj_pairs_checksum_dev": (C.c_int, [_vp, _vp, _u64, _P(_u64)]),
    "rhj_generate_dev": (C.c_int, [_vp, C.c_int, _vp, _u64, _u64, _u64, _u64, C.c_int]),
    "rhj_expected_pkfk_dev": (C.c_int, [_vp, _vp, _u64, _P(_u64), _P(_u64)]),
    "rhj_remap_keys_dev": (C.c_int, [_vp, _vp, _u64, C.c_int, _u64]),
    "rhj_col_filter": (C.c_int, [_vp, _vp, _vp, _u64, C.c_int, _u64, _vp, _P(_u64)]),
    "rhj_gather_tuples": (C.c_int, [_vp, _vp, _vp, _u64, C.c_int, _vp]),
    "rhj_pairs_split": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "rhj_gather_u64": (C.c_int, [_vp, _vp, _vp, _u64, _vp]),
    "rhj_rows_filter_equal": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _P(_u64)]),
    "rhj_sum_gather": (C.c_int, [_vp, _vp, _vp, _u64, _P(_u64)]),
    "rhj_dev_alloc": (C.c_int, [_vp, _u64, _P(_vp)]),
    "rhj_dev_free": (C.c_int, [_vp, _vp]),
    "rhj_copy_h2d": (C.c_int, [_vp, _vp, _vp, _u64]),
    "rhj_copy_d2h": (C.c_int, [_vp, _vp, _vp, _u64]),
    "rhj_dev_mem_info": (C.c_int, [_vp, _P(_u64), _P(_u64)]),
}


def load_library():
    """Load librhj_hip.so and declare every prototype.  Raises if the library is not built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `make -C radixhashjoin_amd/csrc` "
                              f"(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)         # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def plan(nR, nS, opts=None):
    """Resolved radix plan for these sizes (host logic only, works without a GPU)."""
    lib = load_library()
    out = Opts()
    rc = lib.rhj_plan(nR, nS, C.byref(opts) if opts is not None else None, C.byref(out))
    if rc != RHJ_OK:
        raise RhjError(rc, "bad options")
    return out


def mix64(x):
    """numpy form of rhj_mix64 (include/rhj.h): the bijection whose bits the engine's joins take their radix digits and
    owner classes from (splitmix64's finaliser).  x: uint64 array or scalar."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unmix64(h):
    """inverse of mix64: unmix64(mix64(x)) == x (tests craft inputs whose MIXED value has a chosen bit pattern)"""
    with np.errstate(over="ignore"):
        x = np.asarray(h, dtype=np.uint64)
        x = x ^ (x >> np.uint64(31)) ^ (x >> np.uint64(62))
        x = x * np.uint64(0x319642B2D24D8EC3)
        x = x ^ (x >> np.uint64(27)) ^ (x >> np.uint64(54))
        x = x * np.uint64(0x96DE1B173F119089)
        x = x ^ (x >> np.uint64(30)) ^ (x >> np.uint64(60))
        return x - np.uint64(0x9E3779B97F4A7C15)


def narrow_key_offset(n):
    """byte offset of the rowID array in a narrow buffer of n tuples (include/rhj.h, multi-GPU wire format)"""
    return load_library().rhj_narrow_key_offset(n)


def narrow_bytes(n):
    return load_library().rhj_narrow_bytes(n)


SHARD_TAGGED, SHARD_GLOBAL16, SHARD_PLAIN = 1, 2, 3      # include/rhj.h: how the receiver restores global rowIDs


def shard_plan(nR, nS, opts=None):
    """(mode, plan): mode = SHARD_TAGGED / SHARD_GLOBAL16 (or SHARD_PLAIN for the 17-18-bit local plans of receivers beyond
    1.1 * 10^9 tuples, which cannot restore rowIDs) when the narrow sharded path (rhj_shard_*) serves a local join of these sizes
    under `plan`, 0 when it does not (exchange 16-byte tuples instead).  The host may use SHARD_PLAIN instead of the other two
    whenever every rowID of both relations is below 2^32."""
    lib = load_library()
    out = Opts()
    rc = lib.rhj_shard_plan(nR, nS, C.byref(opts) if opts is not None else None, C.byref(out))
    if rc < 0:
        raise RhjError(rc, "bad options")
    return rc, out


def _addr(x):
    """device address of: int, DeviceBuffer, or anything with data_ptr() (torch tensor)"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError(f"not a device pointer: {type(x)}")


class DeviceBuffer:
    """A raw HBM allocation owned through the C-ABI (rhj_dev_alloc / rhj_dev_free)."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = _vp()
        engine._chk(engine.lib.rhj_dev_alloc(engine.ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, engine, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(engine, max(arr.nbytes, 16))
        if arr.nbytes:
            engine._chk(engine.lib.rhj_copy_h2d(engine.ctx, b.ptr, arr.ctypes.data, arr.nbytes))
        return b

    def to_numpy(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            self.engine._chk(self.engine.lib.rhj_copy_d2h(self.engine.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.engine.lib.rhj_dev_free(self.engine.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_libc = C.CDLL(None)
_libc.free.argtypes = [_vp]
_libc_free = _libc.free


class Engine:
    """One rhj_ctx: a HIP stream + HBM workspace on one GPU.  Not thread-safe (one per caller thread)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.ctx = None
        c = _vp()
        rc = self.lib.rhj_init(device, C.byref(c))
        if rc != RHJ_OK:
            raise RhjError(rc, (self.lib.rhj_last_error(None) or b"").decode())
        self.ctx = c

    def close(self):
        if self.ctx:
            self.lib.rhj_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != RHJ_OK and rc not in allow:
            raise RhjError(rc, (self.lib.rhj_last_error(self.ctx) or b"").decode())
        return rc

    # ---- context ------------------------------------------------------------------------------
    def set_stream(self, raw_stream):
        self._chk(self.lib.rhj_set_stream(self.ctx, raw_stream))

    def set_option(self, name, value):
        """tuning / test knobs of include/rhj.h (results never depend on them)"""
        self._chk(self.lib.rhj_set_option(self.ctx, name.encode(), int(value)))

    def info(self, name):
        """what the last join did ("last.narrow", "last.join_kernel"; include/rhj.h)"""
        v = C.c_int64(0)
        self._chk(self.lib.rhj_get_info(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def set_profiling(self, on=True):
        """True / 1: time every launch of a call; 2: accumulate the launches of successive calls (see rhj.h); False / 0: off"""
        self._chk(self.lib.rhj_set_profiling(self.ctx, 2 if on == 2 and on is not True else 1 if on else 0))

    def timings(self):
        t = Timings()
        self._chk(self.lib.rhj_get_timings(self.ctx, C.byref(t)))
        return t.as_dict()

    def launch_timings(self, capacity=4096):
        """[(kind name, ms)] of every timed launch span of the last call, in launch order (profiling must be on)"""
        kinds, ms, n = (_i32 * capacity)(), (C.c_double * capacity)(), C.c_uint32()
        self._chk(self.lib.rhj_get_launch_timings(self.ctx, kinds, ms, capacity, C.byref(n)))
        return [(KERNEL_KINDS[kinds[i]], ms[i]) for i in range(min(n.value, capacity))]

    def sync(self):
        self._chk(self.lib.rhj_sync(self.ctx))

    def reserve(self, nR, nS, opts=None):
        self._chk(self.lib.rhj_reserve(self.ctx, nR, nS, C.byref(opts) if opts is not None else None))

    def release_workspace(self):
        self._chk(self.lib.rhj_release_workspace(self.ctx))

    def mem_info(self):
        f, t = _u64(), _u64()
        self._chk(self.lib.rhj_dev_mem_info(self.ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        return DeviceBuffer.from_numpy(self, arr)

    # ---- the drop-in (host arrays) -----------------------------------------------------------
    def join(self, R, S, opts=None):
        """rhj_join: host AoS in -> numpy array of (rowR,rowS) pairs (copied out of the result page)."""
        R = np.ascontiguousarray(R, dtype=TUPLE)
        S = np.ascontiguousarray(S, dtype=TUPLE)
        page, n = _vp(), _u64()
        self._chk(self.lib.rhj_join(self.ctx, R.ctypes.data, len(R), S.ctypes.data, len(S),
                                    C.byref(opts) if opts is not None else None, C.byref(page), C.byref(n)))
        out = np.empty(n.value, dtype=PAIR)
        if page.value:
            head = C.c_uint64.from_address(page.value).value       # bucket_info::next must be NULL
            assert head == 0
            C.memmove(out.ctypes.data, page.value + 8, out.nbytes)
            C.CDLL(None).free(_vp(page.value))
        else:
            assert n.value == 0
        return out

    def join_batch(self, joins, keep_pairs=True, timed=False):
        """rhj_join_batch over a list of (R, S) host relations: list of pair arrays (or of counts with keep_pairs=False: the pages
        are then freed right away, as ~Result does); timed: also the seconds spent inside the C call"""
        import time
        rel = [(np.ascontiguousarray(R, dtype=TUPLE), np.ascontiguousarray(S, dtype=TUPLE)) for R, S in joins]
        n = len(rel)
        desc = (JoinDesc * max(n, 1))()
        for i, (R, S) in enumerate(rel):
            desc[i] = JoinDesc(R.ctypes.data, len(R), S.ctypes.data, len(S))
        pages, counts = (_vp * max(n, 1))(), (_u64 * max(n, 1))()
        t0 = time.perf_counter()
        rc = self.lib.rhj_join_batch(self.ctx, n, desc, pages, counts)
        dt = time.perf_counter() - t0
        self._chk(rc)
        out = []
        for i in range(n):
            if keep_pairs:
                a = np.empty(counts[i], dtype=PAIR)
                if pages[i]:
                    assert C.c_uint64.from_address(pages[i]).value == 0          # bucket_info::next
                    C.memmove(a.ctypes.data, pages[i] + 8, a.nbytes)
                else:
                    assert counts[i] == 0
                out.append(a)
            else:
                out.append(int(counts[i]))
            if pages[i]:
                _libc_free(_vp(pages[i]))
        return (out, dt) if timed else out

    def join_count_only_page(self, R, S, opts=None, timed=False):
        """rhj_join exactly as the C++ mirror calls it, the result page freed right away (what ~Result does): for
        timing the drop-in without numpy's copy of the pairs.  timed=True also returns the seconds spent in the C call."""
        import time
        R = np.ascontiguousarray(R, dtype=TUPLE)
        S = np.ascontiguousarray(S, dtype=TUPLE)
        page, n = _vp(), _u64()
        t0 = time.perf_counter()
        rc = self.lib.rhj_join(self.ctx, R.ctypes.data, len(R), S.ctypes.data, len(S),
                               C.byref(opts) if opts is not None else None, C.byref(page), C.byref(n))
        dt = time.perf_counter() - t0
        self._chk(rc)
        if page.value:
            _libc_free(_vp(page.value))
        return (n.value, dt) if timed else n.value

    # ---- device-resident ------------------------------------------------------------------------
    def join_dev(self, d_R, nR, d_S, nS, d_out=None, capacity=0, opts=None, allow_overflow=False):
        n = _u64()
        rc = self.lib.rhj_join_dev(self.ctx, _addr(d_R), nR, _addr(d_S), nS,
                                   C.byref(opts) if opts is not None else None, _addr(d_out), capacity, C.byref(n))
        self._chk(rc, allow=(RHJ_E_OVERFLOW,) if allow_overflow else ())
        return n.value

    def histogram(self, d_rel, n, shift, bits, d_hist):
        self._chk(self.lib.rhj_histogram(self.ctx, _addr(d_rel), n, shift, bits, _addr(d_hist)))

    def prefix(self, d_hist, nbins, d_start):
        self._chk(self.lib.rhj_prefix(self.ctx, _addr(d_hist), nbins, _addr(d_start)))

    def partition(self, d_in, n, bits1, bits2, d_out, d_part_start):
        self._chk(self.lib.rhj_partition(self.ctx, _addr(d_in), n, bits1, bits2, _addr(d_out), _addr(d_part_start)))

    def partition_at(self, d_in, n, shift, bits, d_out, d_part_start):
        self._chk(self.lib.rhj_partition_at(self.ctx, _addr(d_in), n, shift, bits, _addr(d_out), _addr(d_part_start)))

    def owner_histogram(self, d_rel, n, shift, bits, d_hist):
        """rhj_histogram on bits of mix64(payload): the multi-GPU owner classes"""
        self._chk(self.lib.rhj_owner_histogram(self.ctx, _addr(d_rel), n, shift, bits, _addr(d_hist)))

    def owner_split(self, d_in, n, shift, bits, d_out, d_class_start):
        """rhj_partition_at on bits of mix64(payload), tuples unchanged"""
        self._chk(self.lib.rhj_owner_split(self.ctx, _addr(d_in), n, shift, bits, _addr(d_out), _addr(d_class_start)))

    def bucket_join(self, d_Rp, d_startR, d_Sp, d_startS, nparts, radix_bits, d_out=None, capacity=0,
                    probe_split=0, allow_overflow=False):
        n = _u64()
        rc = self.lib.rhj_bucket_join(self.ctx, _addr(d_Rp), _addr(d_startR), _addr(d_Sp), _addr(d_startS), nparts,
                                      radix_bits, probe_split, _addr(d_out), capacity, C.byref(n))
        self._chk(rc, allow=(RHJ_E_OVERFLOW,) if allow_overflow else ())
        return n.value

    # ---- multi-GPU stage entry points (include/rhj.h) --------------------------------------
    def shard_stats(self, side, d_rel, n, shift, bits):
        """class histogram (numpy int64 [2^bits]) and rowID range (min, max) of a shard; synchronises"""
        hist = np.zeros(1 << bits, dtype=np.uint64)
        kmin, kmax = _u64(), _u64()
        self._chk(self.lib.rhj_shard_stats(self.ctx, side, _addr(d_rel), n, shift, bits, hist.ctypes.data, C.byref(kmin), C.byref(kmax)))
        return hist.astype(np.int64), kmin.value, kmax.value

    def shard_split(self, side, d_rel, n, shift, bits, key_base, d_narrow_out, d_class_start=None):
        self._chk(self.lib.rhj_shard_split(self.ctx, side, _addr(d_rel), n, shift, bits, key_base, _addr(d_narrow_out),
                                           _addr(d_class_start)))

    def shard_split_peer(self, side, d_rel, n, shift, bits, key_base, owner, dst_class_start, peer_payloads, peer_rowids):
        """rhj_shard_split_peer: owner (uint8[2^bits]) and dst_class_start (uint64[2^bits]) numpy arrays, peer_* lists of device
        pointers / buffers, one per rank"""
        owner = np.ascontiguousarray(owner, dtype=np.uint8)
        dst = np.ascontiguousarray(dst_class_start, dtype=np.uint64)
        nr = len(peer_payloads)
        pp = (_vp * nr)(*[_addr(x) for x in peer_payloads])
        pk = (_vp * nr)(*[_addr(x) for x in peer_rowids])
        self._chk(self.lib.rhj_shard_split_peer(self.ctx, side, _addr(d_rel), n, shift, bits, key_base, owner.ctypes.data, dst.ctypes.data,
                                                pp, pk, nr))

    def shard_partition(self, side, d_payloads, d_rowids, m, seg_off, row0, plan, mode):
        assert len(row0) == len(seg_off) - 1
        seg = (C.c_uint64 * len(seg_off))(*[int(x) for x in seg_off])
        base = (C.c_uint64 * len(row0))(*[int(x) for x in row0])
        self._chk(self.lib.rhj_shard_partition(self.ctx, side, _addr(d_payloads), _addr(d_rowids), m, len(seg_off) - 1, seg, base,
                                               C.byref(plan), mode))

    def shard_join(self, d_out=None, capacity=0, allow_overflow=False):
        n = _u64()
        rc = self.lib.rhj_shard_join(self.ctx, _addr(d_out), capacity, C.byref(n))
        self._chk(rc, allow=(RHJ_E_OVERFLOW,) if allow_overflow else ())
        return n.value

    # ---- query-layer kernels -------------------------------------------------------------
    def col_filter(self, d_col, d_rows_in, n_in, op, value, d_rows_out):
        n = _u64()
        self._chk(self.lib.rhj_col_filter(self.ctx, _addr(d_col), _addr(d_rows_in), n_in, ord(op), value, _addr(d_rows_out), C.byref(n)))
        return n.value

    def gather_tuples(self, d_col, d_rows, n, key_is_position, d_tuples):
        self._chk(self.lib.rhj_gather_tuples(self.ctx, _addr(d_col), _addr(d_rows), n, 1 if key_is_position else 0, _addr(d_tuples)))

    def pairs_split(self, d_pairs, n, d_r, d_s):
        self._chk(self.lib.rhj_pairs_split(self.ctx, _addr(d_pairs), n, _addr(d_r), _addr(d_s)))

    def gather_u64(self, d_src, d_idx, n, d_dst):
        self._chk(self.lib.rhj_gather_u64(self.ctx, _addr(d_src), _addr(d_idx), n, _addr(d_dst)))

    def rows_filter_equal(self, d_colA, d_rowsA, d_colB, d_rowsB, n, d_pos_out):
        m = _u64()
        self._chk(self.lib.rhj_rows_filter_equal(self.ctx, _addr(d_colA), _addr(d_rowsA), _addr(d_colB), _addr(d_rowsB), n,
                                                 _addr(d_pos_out), C.byref(m)))
        return m.value

    def sum_gather(self, d_col, d_rows, n):
        s = _u64()
        self._chk(self.lib.rhj_sum_gather(self.ctx, _addr(d_col), _addr(d_rows), n, C.byref(s)))
        return s.value

    def pairs_checksum(self, d_pairs, n):
        c = _u64()
        self._chk(self.lib.rhj_pairs_checksum_dev(self.ctx, _addr(d_pairs), n, C.byref(c)))
        return c.value

    def generate(self, kind, d_out, n, row0=0, D=1, seed=0, theta_milli=0):
        self._chk(self.lib.rhj_generate_dev(self.ctx, kind, _addr(d_out), n, row0, D, seed, theta_milli))

    def remap_keys(self, d_rel, n, shift, add=0):
        """payload = (k << shift) + add for generated payloads mix(k), in place (same PK/FK pair set, aligned / dense join values)"""
        self._chk(self.lib.rhj_remap_keys_dev(self.ctx, _addr(d_rel), n, shift, add))

    def expected_pkfk(self, d_S, n):
        cnt, c = _u64(), _u64()
        self._chk(self.lib.rhj_expected_pkfk_dev(self.ctx, _addr(d_S), n, C.byref(cnt), C.byref(c)))
        return cnt.value, c.value


GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF, GEN_S_DISJOINT, GEN_CONST = 0, 1, 2, 3, 4
