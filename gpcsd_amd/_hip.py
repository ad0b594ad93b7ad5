"""ctypes binding of libgpcsd_hip.so (include/gpcsd_hip.h).  No CPU fallback: if the HIP library is missing
or no GPU is visible, every compute entry point raises."""
import ctypes
import os
import threading
import weakref

import numpy as np

# A context uses four streams; a process that also runs torch / RCCL streams should give the HIP runtime more than its
# default 4 hardware queues, or streams that share a queue serialise.  Only effective if HIP is not initialised yet.
_HWQ_WANTED = 16          # hardware queues asked of the runtime when the caller set nothing.  8 keep ONE context's four streams apart from
                          # torch's and RCCL's; a SECOND live context (a generator model beside the fitted one, as the reference's
                          # simulation scripts have them) then lands on shared queues and its step runs 1.5 x slower (cfg3: 1.17 against
                          # 0.79 ms per paired step with 8, 0.79 with 16 -- tools/stall_probe.py cfg3 4)


def _hip_runtime_already_initialised():
    """True when this process has initialised HIP through torch before importing gpcsd_amd (the only runtime client this
    package knows how to ask); the environment variable is read by the runtime once, at initialisation."""
    import sys
    torch = sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


def _ensure_hw_queues():
    import warnings
    have = os.environ.get("GPU_MAX_HW_QUEUES")
    if have is None:
        os.environ["GPU_MAX_HW_QUEUES"] = str(_HWQ_WANTED)
        if _hip_runtime_already_initialised():
            warnings.warn("gpcsd_amd was imported after the HIP runtime was initialised with its default of 4 hardware queues: "
                          "a context's four streams plus torch's / RCCL's will share queues and serialise (a multi-rank step "
                          "was measured at 3.7 instead of 1.94 ms).  Import gpcsd_amd -- or set GPU_MAX_HW_QUEUES=%d -- before "
                          "the first CUDA/HIP call." % _HWQ_WANTED, RuntimeWarning, stacklevel=3)
        return
    try:
        n = int(have)
    except ValueError:
        return
    if n < 8:
        warnings.warn("GPU_MAX_HW_QUEUES=%d: gpcsd_amd uses four streams per context beside torch's and RCCL's; with fewer than "
                      "%d hardware queues they share queues and serialise (DESIGN 4.8)." % (n, 8), RuntimeWarning,
                      stacklevel=3)


_ensure_hw_queues()

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPCSD_LIB_PATH: developer knob for A/B timing of two builds in one session (tools/ab_bench.py); the default is the in-tree build
LIB_PATH = os.environ.get("GPCSD_LIB_PATH") or os.path.join(_HERE, "libgpcsd_hip.so")

MAX_TEMPORAL = 8
KIND_SE, KIND_MATERN, KIND_HOST = 0, 1, 2
ERR_CAPACITY = -7
MAX_EIG_N = 4096                 # GPCSD_MAX_EIG_N: rows of one eigenproblem (after symmetry folding)
MAX_GEMM_LD_KMAJOR = 1 << 23     # GPCSD_MAX_GEMM_LD_KMAJOR: ntrials * nt of one resident block of trials
PRED_CSD, PRED_LFP, PRED_BOTH = 1, 2, 3

_c_double_p = ctypes.POINTER(ctypes.c_double)


class HParams(ctypes.Structure):
    """struct gpcsd_hparams"""
    _fields_ = [("R", ctypes.c_double), ("eps", ctypes.c_double), ("ell_s", ctypes.c_double * 2),
                ("n_temporal", ctypes.c_int), ("kind", ctypes.c_int * MAX_TEMPORAL),
                ("ell_t", ctypes.c_double * MAX_TEMPORAL), ("sigma2_t", ctypes.c_double * MAX_TEMPORAL),
                ("n_sig2n", ctypes.c_int), ("sig2n", _c_double_p), ("jitter", ctypes.c_double)]


# the same struct as a NumPy record: a whole batch of hyper-parameter sets is filled column by column
HPARAMS_DTYPE = np.dtype({"names": ["R", "eps", "ell_s", "n_temporal", "kind", "ell_t", "sigma2_t", "n_sig2n", "sig2n", "jitter"],
                          "formats": ["f8", "f8", ("f8", (2,)), "i4", ("i4", (MAX_TEMPORAL,)), ("f8", (MAX_TEMPORAL,)),
                                      ("f8", (MAX_TEMPORAL,)), "i4", "u8", "f8"],
                          "offsets": [getattr(HParams, n).offset for n, _ in HParams._fields_],
                          "itemsize": ctypes.sizeof(HParams)})


class HParamsBatch:
    """B hyper-parameter structs in one contiguous block (what gpcsd_loglik_grad_batch takes), with the noise variances they
    point to (one per set, or one list of nx per set); built from arrays without a Python loop over the sets."""

    def __init__(self, R, eps, ell_s, kinds, ell_t, sigma2_t, sig2n, jitter):
        R = np.asarray(R, dtype=np.float64)
        B = R.shape[0]
        C = len(kinds)
        if not (1 <= C <= MAX_TEMPORAL):
            raise ValueError("between 1 and %d temporal covariance components are supported" % MAX_TEMPORAL)
        rec = np.zeros(B, dtype=HPARAMS_DTYPE)
        rec["R"] = R
        rec["eps"] = eps if eps is not None else 0.0
        ell_s = np.asarray(ell_s, dtype=np.float64).reshape(B, -1)
        rec["ell_s"][:, :ell_s.shape[1]] = ell_s
        rec["n_temporal"] = C
        rec["kind"][:, :C] = np.asarray(kinds, dtype=np.int32)[None, :]
        rec["ell_t"][:, :C] = np.asarray(ell_t, dtype=np.float64).reshape(B, C)
        rec["sigma2_t"][:, :C] = np.asarray(sigma2_t, dtype=np.float64).reshape(B, C)
        # scalar noise: sig2n (B,); per-electrode lists: sig2n (B, nx) -- each struct points at its own row
        self.sig = np.ascontiguousarray(np.asarray(sig2n, dtype=np.float64).reshape(B, -1))
        nsig = self.sig.shape[1]
        if nsig == 1:
            self.sig = self.sig.reshape(B)                   # (scalar noise: one value per set, as before)
        rec["n_sig2n"] = nsig
        rec["sig2n"] = self.sig.ctypes.data + 8 * nsig * np.arange(B, dtype=np.uint64)
        rec["jitter"] = jitter
        self.rec = rec
        self.B = B

    def __len__(self):
        return self.B

    def pointer(self):
        return ctypes.cast(self.rec.ctypes.data, ctypes.POINTER(HParams))


class HipUnavailable(RuntimeError):
    pass


class GPCSDCapacityError(RuntimeError):
    """A problem exceeds a capacity limit of this build (include/gpcsd_hip.h: GPCSD_MAX_EIG_N, GPCSD_MAX_GEMM_LD_KMAJOR).
    Deliberately not a ValueError / LinAlgError: fit() must not treat it as a failed restart and carry on."""


_lib = None
_lib_lock = threading.Lock()

# name -> (restype, argtypes); every symbol include/gpcsd_hip.h declares
_P = ctypes.c_void_p
_D = ctypes.c_double
_I = ctypes.c_int
_L = ctypes.c_long
_DP = _c_double_p
SIGNATURES = {
    "gpcsd_ctx_create": (_I, [_I, ctypes.POINTER(_P)]),
    "gpcsd_ctx_destroy": (_I, [_P]),
    "gpcsd_ctx_stream_handle": (_I, [_P, _I, ctypes.POINTER(ctypes.c_ulonglong)]),
    "gpcsd_last_error": (ctypes.c_char_p, [_P]),
    "gpcsd_version": (_I, []),
    "gpcsd_device_synchronize": (_I, [_P]),
    "gpcsd_host_alloc": (_I, [ctypes.c_size_t, ctypes.POINTER(_P)]),
    "gpcsd_host_free": (_I, [_P]),
    "gpcsd_device_pci_bus_id": (_I, [_I, ctypes.c_char_p, _I]),
    "gpcsd_set_lfp": (_I, [_P, _DP, _I, _I, _I]),
    "gpcsd_set_geometry_1d": (_I, [_P, _DP, _I, _DP, _DP, _I]),
    "gpcsd_set_geometry_2d": (_I, [_P, _DP, _I, _DP, _DP, _I, _DP, _DP, _I]),
    "gpcsd_set_time": (_I, [_P, _DP, _I]),
    "gpcsd_set_host_temporal_gram": (_I, [_P, _DP, _I, _DP, _I, _I]),
    "gpcsd_set_host_temporal_dgram": (_I, [_P, _DP, _I, _I]),
    "gpcsd_b_fwd_1d": (_I, [_P, _DP, _L, _D, _DP]),
    "gpcsd_trad_csd": (_I, [_P, _DP, _L, _L, _L, _I, _DP]),
    "gpcsd_b_fwd_2d": (_I, [_P, _DP, _DP, _DP, _L, _D, _D, _DP]),
    "gpcsd_gram_temporal": (_I, [_P, _I, _DP, _I, _DP, _I, _D, _D, _DP]),
    "gpcsd_ks_csd_1d": (_I, [_P, _DP, _I, _D, _DP]),
    "gpcsd_ks_csd_2d": (_I, [_P, _DP, _I, _D, _D, _DP]),
    "gpcsd_kphi_1d": (_I, [_P, _DP, _I, _DP, _DP, _I, _D, _D, _DP, _I, _DP]),
    "gpcsd_kphig_1d": (_I, [_P, _DP, _I, _DP, _DP, _I, _DP, _I, _D, _D, _DP]),
    "gpcsd_kphi_2d": (_I, [_P, _DP, _I, _DP, _DP, _I, _DP, _DP, _I, _D, _D, _D, _D, _DP, _I, _DP]),
    "gpcsd_kphig_2d": (_I, [_P, _DP, _I, _DP, _DP, _I, _DP, _DP, _I, _DP, _I, _D, _D, _D, _D, _DP]),
    "gpcsd_eigh": (_I, [_P, _DP, _I, _DP, _DP]),
    "gpcsd_eigh_psd": (_I, [_P, _DP, _I, _DP, _DP]),
    "gpcsd_eigh_batch": (_I, [_P, _DP, _I, _I, _DP, _DP, ctypes.POINTER(_I)]),
    "gpcsd_eig_D": (_I, [_P, _DP, _I, _DP, _I, _DP, _I, _DP, _DP, _DP]),
    "gpcsd_whitened_quad": (_I, [_P, _DP, _I, _DP, _I, _DP, _DP, _I, _DP]),
    "gpcsd_debug_sytrd": (_I, [_P, _DP, _I, _DP, _DP, _DP, _DP]),
    "gpcsd_debug_stedc": (_I, [_P, _DP, _DP, _I, _DP, _DP]),
    "gpcsd_potrf": (_I, [_P, _DP, _I, _DP]),
    "gpcsd_logdet_chol": (_I, [_P, _DP, _I, _DP]),
    "gpcsd_trsm_lower": (_I, [_P, _DP, _I, _DP, _I, _DP]),
    "gpcsd_gemm": (_I, [_P, _I, _I, _I, _I, _I, _DP, _DP, _DP]),
    "gpcsd_loglik_dense_chol": (_I, [_P, _DP, _I, _DP, _I, _D, _DP, _I, _DP]),
    "gpcsd_loglik": (_I, [_P, ctypes.POINTER(HParams), _DP]),
    "gpcsd_loglik_parts": (_I, [_P, ctypes.POINTER(HParams), _DP]),
    "gpcsd_loglik_parts_async": (_I, [_P, ctypes.POINTER(HParams)]),
    "gpcsd_loglik_parts_wait": (_I, [_P, _DP]),
    "gpcsd_loglik_predict_async": (_I, [_P, ctypes.POINTER(HParams), ctypes.POINTER(HParams), _DP, _I, _DP, _I, _I, _I]),
    "gpcsd_prefetch_pair": (_I, [_P, ctypes.POINTER(HParams), ctypes.POINTER(HParams), _DP, _I, _DP, _I]),
    "gpcsd_prefetch_stats": (_I, [_P, ctypes.POINTER(_L), ctypes.POINTER(_L)]),
    "gpcsd_loglik_grad": (_I, [_P, ctypes.POINTER(HParams), _DP, _DP, _I]),
    "gpcsd_loglik_grad_batch": (_I, [_P, ctypes.POINTER(HParams), _I, _DP, _DP, _I, ctypes.POINTER(_I)]),
    "gpcsd_predict": (_I, [_P, ctypes.POINTER(HParams), _DP, _I, _DP, _I, _I, _DP, _DP, _DP, _DP]),
    "gpcsd_predict_resident": (_I, [_P, ctypes.POINTER(HParams), _DP, _I, _DP, _I, _I, _I]),
    "gpcsd_fetch": (_I, [_P, ctypes.c_char_p, _DP, _L]),
    "gpcsd_device_buffer": (_I, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]),
    "gpcsd_sample_prior": (_I, [_P, ctypes.POINTER(HParams), _I, _DP, _I, _DP]),
    "gpcsd_set_gram_precision": (_I, [_P, _I]),
    "gpcsd_fold_gemm": (_I, [_P, _I, ctypes.POINTER(_L)]),
    "gpcsd_ll_tridiag": (_I, [_P, _I, ctypes.POINTER(_L)]),
    "gpcsd_pair_share_x": (_I, [_P, _I, ctypes.POINTER(_L)]),
    "gpcsd_pair_share_s": (_I, [_P, _I, ctypes.POINTER(_L)]),
    "gpcsd_q_pipeline": (_I, [_P, _I, ctypes.POINTER(_L)]),
    "gpcsd_bounce_stats": (_I, [_P, ctypes.POINTER(_L)]),
    "gpcsd_q_pipeline_stats": (_I, [_P, ctypes.POINTER(_I), ctypes.POINTER(_L), ctypes.c_longlong]),
    "gpcsd_predict_chunked_copy": (_I, [_P, _I, ctypes.POINTER(_L)]),
    "gpcsd_tail_early_exit": (_I, [_P, _I, ctypes.POINTER(_I)]),
    "gpcsd_debug_fault_stage2": (_I, [_P, _I]),
    "gpcsd_decomposition_cache": (_I, [_P, _I, ctypes.POINTER(_L)]),
    "gpcsd_shard_block": (_I, [_I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "gpcsd_combine_loglik": (_I, [_I, _D, _D, _DP]),
    "gpcsd_dist_create": (_I, [_I, ctypes.POINTER(_I), ctypes.POINTER(_P)]),
    "gpcsd_dist_destroy": (_I, [_P]),
    "gpcsd_dist_size": (_I, [_P]),
    "gpcsd_dist_ctx": (_I, [_P, _I, ctypes.POINTER(_P)]),
    "gpcsd_dist_last_error": (ctypes.c_char_p, [_P]),
    "gpcsd_dist_set_geometry_1d": (_I, [_P, _DP, _I, _DP, _DP, _I]),
    "gpcsd_dist_set_geometry_2d": (_I, [_P, _DP, _I, _DP, _DP, _I, _DP, _DP, _I]),
    "gpcsd_dist_set_time": (_I, [_P, _DP, _I]),
    "gpcsd_dist_set_lfp": (_I, [_P, _DP, _I, _I, _I, _I]),
    "gpcsd_dist_loglik": (_I, [_P, ctypes.POINTER(HParams), _DP]),
    "gpcsd_dist_loglik_grad": (_I, [_P, ctypes.POINTER(HParams), _DP, _DP, _I]),
    "gpcsd_dist_loglik_grad_batch": (_I, [_P, ctypes.POINTER(HParams), _I, _DP, _DP, _I, ctypes.POINTER(_I)]),
    "gpcsd_dist_predict": (_I, [_P, ctypes.POINTER(HParams), _DP, _I, _DP, _I, _I, _DP, _DP, _DP, _DP]),
    "gpcsd_prof_enable": (_I, [_P, _I]),
    "gpcsd_prof_tail_clock": (_I, [_P, _I, _DP, ctypes.POINTER(ctypes.c_int), _DP]),
    "gpcsd_prof_reset": (_I, [_P]),
    "gpcsd_prof_get": (_I, [_P, ctypes.c_char_p, _DP, ctypes.POINTER(_L), _DP]),
    "gpcsd_prof_names": (_I, [_P, ctypes.c_char_p, _I]),
    "gpcsd_mfma_f64_peak": (_I, [_P, _DP]),
    "gpcsd_hbm_copy_peak": (_I, [_P, _L, _DP]),
    "gpcsd_gemm_bench": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _DP]),
    "gpcsd_potrf_bench": (_I, [_P, _I, _I, _DP]),
    "gpcsd_potrf_gate_timeouts": (_I, [_P, ctypes.POINTER(_L)]),
    "gpcsd_potrf_diag_probe": (_I, [_P, _DP]),
}


def load_library():
    """dlopen the in-tree HIP library and declare every prototype.  Raises HipUnavailable if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable("%s not found: build it with `python -m gpcsd_amd.build` "
                                 "(gpcsd_amd has no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _parse_cpulist(text):
    """'0-3,8,10-11' (the kernel's cpulist format) -> set of ints."""
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


def bind_host_to_device_numa(device=0):
    """Keep every thread of this process on the CPUs of the NUMA node device `device` hangs off (sched_setaffinity on all
    threads, intersected with the affinity the process already has).  Returns {"node", "cpus", "pci"}, or None when the node
    cannot be told (no sysfs, one node only, no permission) -- then nothing is changed.

    Why: on a two-socket host a process the scheduler may move between the sockets runs the latency-bound queued step of
    bench.py at 1.12 .. 1.25 ms from one run to the next; bound to one socket's CPUs it stays at 1.13 (either socket: it is
    the migration that costs; `taskset -c <node cpulist>` from outside does the same).  Opt-in -- a library does not change
    its host's affinity on its own: bench.py calls it, a multi-rank launcher calls it per rank with LOCAL_RANK."""
    try:
        lib = load_library()
        buf = ctypes.create_string_buffer(64)
        if lib.gpcsd_device_pci_bus_id(int(device), buf, 64) != 0:
            return None
        pci = buf.value.decode().lower()
        nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
        if len(nodes) < 2:
            return None
        node = -1
        try:
            node = int(open("/sys/bus/pci/devices/%s/numa_node" % pci).read())
        except (OSError, ValueError):
            pass
        lists = {nd: _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % nd).read()) for nd in nodes}
        if node not in lists:                      # unknown to the kernel: the node most of the present affinity lies on
            have = os.sched_getaffinity(0)
            node = max(nodes, key=lambda nd: len(lists[nd] & have))
        cpus = lists[node] & os.sched_getaffinity(0)
        if not cpus:
            return None
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except (OSError, ValueError):
                pass
        return {"node": node, "cpus": len(cpus), "pci": pci}
    except (OSError, HipUnavailable):
        return None


def _arr(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


def _ptr(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


class PinnedPool:
    """Recycled page-locked result arrays.  predict() returns host arrays (gpcsd2d.py:328-334) of up to hundreds of MB per
    call; landing them in fresh pageable memory costs a page fault per 4 KiB and a staged copy (~9 GB/s measured).  The pool
    hands out NumPy arrays backed by pinned blocks; a block returns to the pool when the last array viewing it is garbage
    collected (a finalizer on the backing buffer), so results a caller keeps alive are never overwritten -- a caller that
    replaces them call after call (the usual loop) ping-pongs between two blocks and allocates nothing."""
    MIN_BYTES = 1 << 20          # smaller results are not worth a pinned block
    MAX_FREE_PER_SIZE = 3

    def __init__(self):
        self._free = {}
        self._lock = threading.Lock()

    def empty(self, shape):
        n = int(np.prod(shape))
        nbytes = 8 * n
        if nbytes < self.MIN_BYTES:
            return np.empty(shape)
        lib = load_library()
        with self._lock:
            lst = self._free.get(nbytes)
            ptr = lst.pop() if lst else None
        if ptr is None:
            h = ctypes.c_void_p()
            if lib.gpcsd_host_alloc(nbytes, ctypes.byref(h)) != 0 or not h.value:
                return np.empty(shape)                       # pinning refused (limits): pageable still works, slower
            ptr = h.value
        buf = (ctypes.c_double * n).from_address(ptr)
        weakref.finalize(buf, self._release, ptr, nbytes)
        return np.frombuffer(buf, dtype=np.float64).reshape(shape)

    def _release(self, ptr, nbytes):
        with self._lock:
            lst = self._free.setdefault(nbytes, [])
            if len(lst) < self.MAX_FREE_PER_SIZE:
                lst.append(ptr)
                return
        try:
            load_library().gpcsd_host_free(ctypes.c_void_p(ptr))
        except Exception:
            pass


pinned_pool = PinnedPool()


class DeviceArray:
    """A float64 C-contiguous view of library-owned device memory (`__cuda_array_interface__` version 3)."""

    def __init__(self, ptr, shape, owner, name=None):
        self.ptr, self.shape, self._owner, self.name = int(ptr), tuple(shape), owner, name   # (the context stays alive with the view)
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 3,
                                         "strides": None, "stream": None}

    def __getitem__(self, i):
        """Sub-array i along the first axis (component i of a `*_list` buffer), still a zero-copy device view."""
        i = int(i)
        if not self.shape or not (0 <= i < self.shape[0]):
            raise IndexError(i)
        sub = self.shape[1:]
        return DeviceArray(self.ptr + 8 * i * int(np.prod(sub)), sub, self._owner, None)

    def torch(self, device=None):
        import torch
        return torch.as_tensor(self, device=device if device is not None else "cuda")

    def numpy(self):
        """A host copy (through the library's page-locked pool when the view is a whole named buffer)."""
        if self.name is not None:
            return self._owner.fetch(self.name, self.shape)
        return self.torch().cpu().numpy()


class Context:
    """One device + stream + resident LFP / geometry (gpcsd_ctx)."""

    def __init__(self, device=None):
        lib = load_library()
        if device is None:
            device = int(os.environ.get("GPCSD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = ctypes.c_void_p()
        rc = lib.gpcsd_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            msg = lib.gpcsd_last_error(None)
            raise HipUnavailable("gpcsd_ctx_create(device=%d) failed (rc=%d): %s -- gpcsd_amd needs an AMD GPU; "
                                 "there is no CPU fallback" % (device, rc, (msg or b"").decode()))
        self._lib = lib
        self._h = h
        self.device = int(device)
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpcsd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream_handles(self):
        """The context's four hipStream_t as integers: (main, temporal chain, spatial chain, side stream)."""
        out = []
        for which in range(4):
            v = ctypes.c_ulonglong()
            self._check(self._lib.gpcsd_ctx_stream_handle(self._h, which, ctypes.byref(v)))
            out.append(int(v.value))
        return tuple(out)

    def stream_pool_stats(self):
        """(stream sets created, stream sets taken over from a closed context) in this process."""
        out = []
        for which in (-1, -2):
            v = ctypes.c_ulonglong()
            self._check(self._lib.gpcsd_ctx_stream_handle(None, which, ctypes.byref(v)))
            out.append(int(v.value))
        return tuple(out)

    # ---- error convention (include/gpcsd_hip.h) ----
    def _check(self, rc):
        if rc == 0:
            return
        msg = (self._lib.gpcsd_last_error(self._h) or b"").decode()
        if rc > 0:
            raise np.linalg.LinAlgError(msg or "numerical failure (status %d)" % rc)
        if rc == ERR_CAPACITY:
            raise GPCSDCapacityError(msg)
        if rc in (-3, -22):
            raise ValueError(msg)
        raise RuntimeError("libgpcsd_hip error %d: %s" % (rc, msg))

    def make_hparams(self, R, eps, ell_s, temporal, sig2n, jitter):
        """temporal: list of (kind, ell, sigma2).  Returns (HParams, keepalive).  Called once or twice per evaluation on the host's
        critical path (a queued step is a millisecond): plain Python numbers go straight into the struct, NumPy is only asked when
        an argument is an array."""
        hp = HParams()
        hp.R = float(R)
        hp.eps = float(eps) if eps is not None else 0.0
        if isinstance(ell_s, (list, tuple)):
            hp.ell_s[0] = float(ell_s[0])
            hp.ell_s[1] = float(ell_s[1]) if len(ell_s) > 1 else 0.0
        else:
            ell_s = np.atleast_1d(np.asarray(ell_s, dtype=np.float64))
            hp.ell_s[0] = float(ell_s[0])
            hp.ell_s[1] = float(ell_s[1]) if ell_s.size > 1 else 0.0
        nt = len(temporal)
        if not (1 <= nt <= MAX_TEMPORAL):
            raise ValueError("between 1 and %d temporal covariance components are supported" % MAX_TEMPORAL)
        hp.n_temporal = nt
        kind, ell_t, s2_t = hp.kind, hp.ell_t, hp.sigma2_t
        for i, (k, ell, s2) in enumerate(temporal):
            kind[i] = int(k)
            ell_t[i] = float(ell)
            s2_t[i] = float(s2)
        if isinstance(sig2n, (float, int)) or np.ndim(sig2n) == 0:
            sig = (ctypes.c_double * 1)(float(sig2n))
            hp.n_sig2n = 1
            hp.sig2n = ctypes.cast(sig, _c_double_p)
        else:
            sig = np.ascontiguousarray(np.atleast_1d(np.asarray(sig2n, dtype=np.float64)))
            hp.n_sig2n = int(sig.size)
            hp.sig2n = _ptr(sig)
        hp.jitter = float(jitter)
        return hp, sig

    # ---- resident data ----
    def set_lfp(self, lfp):
        lfp = _arr(np.atleast_3d(lfp))
        nx, nt, R = lfp.shape
        self._check(self._lib.gpcsd_set_lfp(self._h, _ptr(lfp), nx, nt, R))

    def set_geometry_1d(self, x, gl_x, gl_w):
        x = _arr(x).reshape(-1)
        gl_x, gl_w = _arr(gl_x).reshape(-1), _arr(gl_w).reshape(-1)
        self._check(self._lib.gpcsd_set_geometry_1d(self._h, _ptr(x), x.size, _ptr(gl_x), _ptr(gl_w), gl_x.size))

    def set_geometry_2d(self, xy, gl_x1, gl_w1, gl_x2, gl_w2):
        xy = _arr(xy)
        gl_x1, gl_w1, gl_x2, gl_w2 = (_arr(v).reshape(-1) for v in (gl_x1, gl_w1, gl_x2, gl_w2))
        self._check(self._lib.gpcsd_set_geometry_2d(self._h, _ptr(xy), xy.shape[0], _ptr(gl_x1), _ptr(gl_w1), gl_x1.size,
                                                    _ptr(gl_x2), _ptr(gl_w2), gl_x2.size))

    def set_time(self, t):
        t = _arr(t).reshape(-1)
        self._check(self._lib.gpcsd_set_time(self._h, _ptr(t), t.size))

    def set_host_temporal_gram(self, Kt, Kt_cross=None):
        """Temporal Gram matrices of user-defined covariances, evaluated by the caller: Kt (nt, nt) summed over components
        and optionally Kt_cross (C, ntstar, nt) per component for predict.  Kt=None returns to the built-in builders."""
        if Kt is None:
            self._check(self._lib.gpcsd_set_host_temporal_gram(self._h, None, 0, None, 0, 0))
            return
        Kt = _arr(Kt)
        nt = Kt.shape[0]
        if Kt.shape != (nt, nt):
            raise ValueError("Kt must be square, got %s" % (Kt.shape,))
        if Kt_cross is None:
            self._check(self._lib.gpcsd_set_host_temporal_gram(self._h, _ptr(Kt), nt, None, 0, 0))
            return
        Kt_cross = _arr(Kt_cross)
        if Kt_cross.ndim != 3 or Kt_cross.shape[2] != nt:
            raise ValueError("Kt_cross must have shape (C, ntstar, %d), got %s" % (nt, Kt_cross.shape))
        self._check(self._lib.gpcsd_set_host_temporal_gram(self._h, _ptr(Kt), nt, _ptr(Kt_cross), Kt_cross.shape[0],
                                                           Kt_cross.shape[1]))

    # ---- operators ----
    def set_host_temporal_dgram(self, dKt):
        """Derivatives of the host temporal Gram handed over last: (2 * n_temporal, nt, nt), ordered (d/d ell_c, d/d sigma2_c)
        per component; None clears."""
        if dKt is None:
            self._check(self._lib.gpcsd_set_host_temporal_dgram(self._h, None, 0, 0))
            return
        dKt = _arr(dKt)
        if dKt.ndim != 3 or dKt.shape[1] != dKt.shape[2]:
            raise ValueError("dKt must have shape (n_matrices, nt, nt)")
        self._check(self._lib.gpcsd_set_host_temporal_dgram(self._h, _ptr(dKt), dKt.shape[1], dKt.shape[0]))

    def b_fwd_1d(self, r, R):
        r = _arr(r)
        out = np.empty_like(r)
        self._check(self._lib.gpcsd_b_fwd_1d(self._h, _ptr(r), r.size, float(R), _ptr(out)))
        return out

    def trad_csd(self, lfp, axis, edge_nan):
        """minus the second difference of lfp along `axis`; the two end positions are -0.0 or NaN (predict_csd.py)"""
        lfp = _arr(lfp)
        out = np.empty_like(lfp)
        n_outer = int(np.prod(lfp.shape[:axis], dtype=np.int64))
        n_inner = int(np.prod(lfp.shape[axis + 1:], dtype=np.int64))
        self._check(self._lib.gpcsd_trad_csd(self._h, _ptr(lfp), n_outer, lfp.shape[axis], n_inner, int(bool(edge_nan)), _ptr(out)))
        return out

    def b_fwd_2d(self, d1, d2, R, eps, w=None):
        if w is not None:
            w = _arr(w)
            out = np.empty_like(w)
            self._check(self._lib.gpcsd_b_fwd_2d(self._h, None, None, _ptr(w), w.size, float(R), float(eps), _ptr(out)))
            return out
        d1, d2 = np.broadcast_arrays(np.asarray(d1, dtype=np.float64), np.asarray(d2, dtype=np.float64))
        d1, d2 = _arr(d1), _arr(d2)
        out = np.empty_like(d1)
        self._check(self._lib.gpcsd_b_fwd_2d(self._h, _ptr(d1), _ptr(d2), None, d1.size, float(R), float(eps), _ptr(out)))
        return out

    def gram_temporal(self, kind, t, tp, ell, sigma2):
        t, tp = _arr(t).reshape(-1), _arr(tp).reshape(-1)
        out = np.empty((t.size, tp.size))
        self._check(self._lib.gpcsd_gram_temporal(self._h, int(kind), _ptr(t), t.size, _ptr(tp), tp.size, float(ell),
                                                  float(sigma2), _ptr(out)))
        return out

    def ks_csd_1d(self, x, ell):
        x = _arr(x).reshape(-1)
        out = np.empty((x.size, x.size))
        self._check(self._lib.gpcsd_ks_csd_1d(self._h, _ptr(x), x.size, float(ell), _ptr(out)))
        return out

    def ks_csd_2d(self, xy, ell1, ell2):
        xy = _arr(xy)
        n = xy.shape[0]
        out = np.empty((n, n))
        self._check(self._lib.gpcsd_ks_csd_2d(self._h, _ptr(xy), n, float(ell1), float(ell2), _ptr(out)))
        return out

    def kphi_1d(self, x, gl_x, gl_w, R, ell, xp=None):
        x, gl_x, gl_w = _arr(x).reshape(-1), _arr(gl_x).reshape(-1), _arr(gl_w).reshape(-1)
        xp_ = None if xp is None else _arr(xp).reshape(-1)
        n2 = x.size if xp_ is None else xp_.size
        out = np.empty((x.size, n2))
        self._check(self._lib.gpcsd_kphi_1d(self._h, _ptr(x), x.size, _ptr(gl_x), _ptr(gl_w), gl_x.size, float(R), float(ell),
                                            _ptr(xp_), 0 if xp_ is None else xp_.size, _ptr(out)))
        return out

    def kphig_1d(self, x, gl_x, gl_w, z, R, ell):
        x, gl_x, gl_w, z = (_arr(v).reshape(-1) for v in (x, gl_x, gl_w, z))
        out = np.empty((x.size, z.size))
        self._check(self._lib.gpcsd_kphig_1d(self._h, _ptr(x), x.size, _ptr(gl_x), _ptr(gl_w), gl_x.size, _ptr(z), z.size,
                                             float(R), float(ell), _ptr(out)))
        return out

    def kphi_2d(self, xy, gl_x1, gl_w1, gl_x2, gl_w2, R, eps, ell1, ell2, xp=None):
        xy = _arr(xy)
        gl_x1, gl_w1, gl_x2, gl_w2 = (_arr(v).reshape(-1) for v in (gl_x1, gl_w1, gl_x2, gl_w2))
        xp_ = None if xp is None else _arr(xp)
        n2 = xy.shape[0] if xp_ is None else xp_.shape[0]
        out = np.empty((xy.shape[0], n2))
        self._check(self._lib.gpcsd_kphi_2d(self._h, _ptr(xy), xy.shape[0], _ptr(gl_x1), _ptr(gl_w1), gl_x1.size, _ptr(gl_x2),
                                            _ptr(gl_w2), gl_x2.size, float(R), float(eps), float(ell1), float(ell2),
                                            _ptr(xp_), 0 if xp_ is None else xp_.shape[0], _ptr(out)))
        return out

    def kphig_2d(self, xy, gl_x1, gl_w1, gl_x2, gl_w2, z, R, eps, ell1, ell2):
        xy, z = _arr(xy), _arr(z)
        gl_x1, gl_w1, gl_x2, gl_w2 = (_arr(v).reshape(-1) for v in (gl_x1, gl_w1, gl_x2, gl_w2))
        out = np.empty((xy.shape[0], z.shape[0]))
        self._check(self._lib.gpcsd_kphig_2d(self._h, _ptr(xy), xy.shape[0], _ptr(gl_x1), _ptr(gl_w1), gl_x1.size, _ptr(gl_x2),
                                             _ptr(gl_w2), gl_x2.size, _ptr(z), z.shape[0], float(R), float(eps), float(ell1),
                                             float(ell2), _ptr(out)))
        return out

    def eigh(self, A, psd=False):
        """numpy.linalg.eigh; psd=True: the caller vouches that A is positive semi-definite (a Gram matrix), which lets orders
        <= 192 take the tridiagonalisation's rank-revealing early exit (tail_early_exit)."""
        A = _arr(A)
        n = A.shape[0]
        if A.shape != (n, n):
            raise ValueError("eigh needs a square matrix")
        w, V = np.empty(n), np.empty((n, n))
        fn = self._lib.gpcsd_eigh_psd if psd else self._lib.gpcsd_eigh
        self._check(fn(self._h, _ptr(A), n, _ptr(w), _ptr(V)))
        return w, V

    def eigh_batch(self, A):
        """A (count, n, n) -> (evals (count, n), evecs (count, n, n), status (count,)) through one shared chain of launches."""
        A = _arr(A)
        if A.ndim != 3 or A.shape[1] != A.shape[2]:
            raise ValueError("eigh_batch needs an array of shape (count, n, n)")
        count, n = A.shape[0], A.shape[1]
        w, V = np.empty((count, n)), np.empty((count, n, n))
        st = np.zeros(count, dtype=np.int32)
        self._check(self._lib.gpcsd_eigh_batch(self._h, _ptr(A), n, count, _ptr(w), _ptr(V),
                                               st.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return w, V, st

    def debug_sytrd(self, A):
        A = _arr(A)
        n = A.shape[0]
        d, e, tau, V = np.empty(n), np.empty(n), np.empty(n), np.empty((n, n))
        self._check(self._lib.gpcsd_debug_sytrd(self._h, _ptr(A), n, _ptr(d), _ptr(e), _ptr(V), _ptr(tau)))
        return d, e[:n - 1], V, tau

    def debug_stedc(self, d, e):
        d, e = _arr(d).reshape(-1), _arr(e).reshape(-1)
        n = d.size
        w, Z = np.empty(n), np.empty((n, n))
        self._check(self._lib.gpcsd_debug_stedc(self._h, _ptr(d), _ptr(e), n, _ptr(w), _ptr(Z)))
        return w, Z

    def eig_D(self, Ks, Kt, sig2n):
        Ks, Kt = _arr(Ks), _arr(Kt)
        nx, nt = Ks.shape[0], Kt.shape[0]
        sig = np.ascontiguousarray(np.atleast_1d(np.asarray(sig2n, dtype=np.float64)))
        Qs, Qt, D = np.empty((nx, nx)), np.empty((nt, nt)), np.empty(nx * nt)
        self._check(self._lib.gpcsd_eig_D(self._h, _ptr(Ks), nx, _ptr(Kt), nt, _ptr(sig), sig.size, _ptr(Qs), _ptr(Qt), _ptr(D)))
        return Qs, Qt, D

    def whitened_quad(self, Qs, Qt, Dvec, resid):
        """sum((Qs^T resid_b Qt)^2 / Dvec) for every trial b of resid (nx, nt, nb) -> (nb,)."""
        Qs, Qt = _arr(Qs), _arr(Qt)
        nx, nt = Qs.shape[0], Qt.shape[0]
        resid = np.ascontiguousarray(np.atleast_3d(np.asarray(resid, dtype=np.float64)))
        if resid.shape[:2] != (nx, nt):
            raise ValueError("resid has shape %s, expected (%d, %d, nb)" % (resid.shape, nx, nt))
        Dv = _arr(np.asarray(Dvec, dtype=np.float64).reshape(-1), (nx * nt,), "Dvec")
        out = np.empty(resid.shape[2])
        self._check(self._lib.gpcsd_whitened_quad(self._h, _ptr(Qs), nx, _ptr(Qt), nt, _ptr(Dv), _ptr(resid), resid.shape[2], _ptr(out)))
        return out

    def potrf(self, A):
        A = _arr(A)
        n = A.shape[0]
        L = np.empty((n, n))
        self._check(self._lib.gpcsd_potrf(self._h, _ptr(A), n, _ptr(L)))
        return L

    def logdet_chol(self, L):
        L = _arr(L)
        out = np.empty(1)
        self._check(self._lib.gpcsd_logdet_chol(self._h, _ptr(L), L.shape[0], _ptr(out)))
        return float(out[0])

    def trsm_lower(self, L, B):
        L, B = _arr(L), _arr(B)
        B2 = B.reshape(L.shape[0], -1)
        X = np.empty_like(B2)
        self._check(self._lib.gpcsd_trsm_lower(self._h, _ptr(L), L.shape[0], _ptr(B2), B2.shape[1], _ptr(X)))
        return X.reshape(B.shape)

    def gemm(self, A, B, transA=False, transB=False):
        A, B = _arr(A), _arr(B)
        M, K = (A.shape[1], A.shape[0]) if transA else A.shape
        N = B.shape[0] if transB else B.shape[1]
        K2 = B.shape[1] if transB else B.shape[0]
        if K != K2:
            raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, K2))
        C = np.empty((M, N))
        self._check(self._lib.gpcsd_gemm(self._h, int(transA), int(transB), M, N, K, _ptr(A), _ptr(B), _ptr(C)))
        return C

    def loglik_dense_chol(self, Ks, Kt, sig2n, lfp):
        Ks, Kt, lfp = _arr(Ks), _arr(Kt), _arr(np.atleast_3d(lfp))
        out = np.empty(1)
        self._check(self._lib.gpcsd_loglik_dense_chol(self._h, _ptr(Ks), Ks.shape[0], _ptr(Kt), Kt.shape[0], float(sig2n),
                                                      _ptr(lfp), lfp.shape[2], _ptr(out)))
        return float(out[0])

    # ---- fused hot calls ----
    def loglik(self, hp):
        out = np.empty(1)
        self._check(self._lib.gpcsd_loglik(self._h, ctypes.byref(hp), _ptr(out)))
        return float(out[0])

    def loglik_parts(self, hp):
        out = np.empty(2)
        self._check(self._lib.gpcsd_loglik_parts(self._h, ctypes.byref(hp), _ptr(out)))
        return float(out[0]), float(out[1])

    def loglik_parts_async(self, hp):
        """Queue loglik_parts(hp) and return; collect with loglik_parts_wait().  Calls queued in between (predict_resident)
        overlap with this evaluation's tail."""
        self._check(self._lib.gpcsd_loglik_parts_async(self._h, ctypes.byref(hp)))

    def loglik_parts_wait(self):
        # (the queued forms are called a thousand times a second: their wrappers keep what does not change from call to call)
        w = getattr(self, "_wait_out", None)
        if w is None:
            out = np.empty(2)
            w = self._wait_out = (out, _ptr(out))
        rc = self._lib.gpcsd_loglik_parts_wait(self._h, w[1])
        if rc:
            self._check(rc)
        return float(w[0][0]), float(w[0][1])

    def loglik_predict_async(self, hp_loglik, hp_predict, z, tstar, type_code, want_lists=True):
        """loglik_parts_async(hp_loglik) + predict_resident(hp_predict, ...) as one queued call with the decompositions of the
        two batched (same bits); collect with loglik_parts_wait() and fetch()."""
        k = getattr(self, "_lpa_args", None)
        if k is None or k[0] is not z or k[1] is not tstar:
            za = _arr(z)
            ta = _arr(tstar).reshape(-1)
            k = (z, tstar, za, ta, _ptr(za), za.shape[0], _ptr(ta), ta.size)
            # the pointers are kept only when they point into the caller's own arrays (no contiguous copy was made): an in-place
            # edit of z / tstar between calls is then seen, as without the cache
            self._lpa_args = k if (za is z and (ta is tstar or ta.base is tstar)) else None
        rc = self._lib.gpcsd_loglik_predict_async(self._h, ctypes.byref(hp_loglik), ctypes.byref(hp_predict), k[4], k[5], k[6], k[7],
                                                  int(type_code), 1 if want_lists else 0)
        if rc:
            self._check(rc)

    def prefetch_pair(self, hp_loglik, hp_predict, z, tstar):
        """Announce the next loglik_predict_async (same arguments): its decomposition chains are queued now and start as soon as
        their streams are free.  True when queued (the paired form applies)."""
        k = getattr(self, "_lpa_args", None)
        if k is None or k[0] is not z or k[1] is not tstar:
            za = _arr(z)
            ta = _arr(tstar).reshape(-1)
            k = (z, tstar, za, ta, _ptr(za), za.shape[0], _ptr(ta), ta.size)
        rc = self._lib.gpcsd_prefetch_pair(self._h, ctypes.byref(hp_loglik), ctypes.byref(hp_predict), k[4], k[5], k[6], k[7])
        if rc < 0:
            self._check(rc)
        return rc == 1

    def potrf_gate_timeouts(self):
        """Gate launches of the dense Cholesky that gave up waiting since the context was created (a performance counter)."""
        n = _L(0)
        self._check(self._lib.gpcsd_potrf_gate_timeouts(self._h, ctypes.byref(n)))
        return int(n.value)

    def prefetch_stats(self):
        """(front halves queued by prefetch_pair, front halves a paired call took over)."""
        q, t = _L(0), _L(0)
        self._check(self._lib.gpcsd_prefetch_stats(self._h, ctypes.byref(q), ctypes.byref(t)))
        return int(q.value), int(t.value)

    def loglik_grad(self, hp, ngrad):
        """(sum log D, local quad, d L_loc / d natural params) with L_loc = -0.5*R_resident*sumlog - 0.5*quad."""
        out = np.empty(2)
        g = np.empty(ngrad)
        self._check(self._lib.gpcsd_loglik_grad(self._h, ctypes.byref(hp), _ptr(out), _ptr(g), int(ngrad)))
        return float(out[0]), float(out[1]), g

    def loglik_grad_batch(self, hps, ngrad):
        """hps: list of HParams (same kernel kinds, the same number of noise entries).  One shared chain of launches for all of them.
        Returns (sumlog (B,), quad (B,), grad (B, ngrad), status (B,)); status[i] > 0: set i failed numerically."""
        B = len(hps)
        arr = hps.pointer() if isinstance(hps, HParamsBatch) else (HParams * B)(*hps)
        out = np.empty((B, 2))
        g = np.empty((B, ngrad))
        st = np.zeros(B, dtype=np.int32)
        self._check(self._lib.gpcsd_loglik_grad_batch(self._h, arr, B, _ptr(out), _ptr(g), int(ngrad),
                                                      st.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return out[:, 0].copy(), out[:, 1].copy(), g, st

    def predict(self, hp, z, tstar, type_code, shape, want_lists=True):
        """shape = (nz, ntstar, ntrials).  Returns dict of arrays for the requested type."""
        z = _arr(z)
        tstar = _arr(tstar).reshape(-1)
        nz, ntstar, R = shape
        C = hp.n_temporal
        res = {}
        bufs = {}
        for name, bit in (("csd", PRED_CSD), ("lfp", PRED_LFP)):
            if type_code & bit:
                bufs[name] = pinned_pool.empty((nz, ntstar, R))
                bufs[name + "_list"] = pinned_pool.empty((C, nz, ntstar, R)) if want_lists else None
            else:
                bufs[name] = None
                bufs[name + "_list"] = None
        self._check(self._lib.gpcsd_predict(self._h, ctypes.byref(hp), _ptr(z), nz, _ptr(tstar), tstar.size, int(type_code),
                                            _ptr(bufs["csd_list"]), _ptr(bufs["csd"]), _ptr(bufs["lfp_list"]), _ptr(bufs["lfp"])))
        for k, v in bufs.items():
            if v is not None:
                res[k] = v
        return res

    def predict_resident(self, hp, z, tstar, type_code, want_lists=True):
        """Compute the posterior mean into device buffers only (no PCIe traffic); read back with fetch()."""
        z = _arr(z)
        tstar = _arr(tstar).reshape(-1)
        self._check(self._lib.gpcsd_predict_resident(self._h, ctypes.byref(hp), _ptr(z), z.shape[0], _ptr(tstar), tstar.size,
                                                     int(type_code), int(bool(want_lists))))

    def fetch(self, name, shape):
        out = pinned_pool.empty(shape)
        self._check(self._lib.gpcsd_fetch(self._h, name.encode(), _ptr(out), out.size))
        return out

    def device_array(self, name, shape):
        """The named device buffer (what fetch() copies out) as a zero-copy object with `__cuda_array_interface__` (float64, C
        order): `torch.as_tensor(ctx.device_array("pred_out_csd", (nz, nt, R)), device="cuda")`, cupy.asarray(...).  The context's
        streams are drained first; the memory belongs to the library and is rewritten by the next call that produces it."""
        p, nb = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        self._check(self._lib.gpcsd_device_buffer(self._h, name.encode(), ctypes.byref(p), ctypes.byref(nb)))
        shape = tuple(int(v) for v in shape)
        if 8 * int(np.prod(shape)) > nb.value:
            raise ValueError("device buffer %r holds %d bytes, shape %r needs %d" % (name, nb.value, shape, 8 * int(np.prod(shape))))
        return DeviceArray(p.value, shape, self, name)

    def sample_prior(self, hp, which, normals):
        normals = _arr(np.atleast_3d(normals))
        out = np.empty_like(normals)
        self._check(self._lib.gpcsd_sample_prior(self._h, ctypes.byref(hp), int(which), _ptr(normals), normals.shape[2], _ptr(out)))
        return out

    # ---- measurement ----
    def synchronize(self):
        self._check(self._lib.gpcsd_device_synchronize(self._h))

    def set_gram_precision(self, bits):
        """64 (default): Gram builders as the reference evaluates them; 32: single-precision Gram build, fp64 everything else."""
        self._check(self._lib.gpcsd_set_gram_precision(self._h, int(bits)))

    def fold_gemm(self, on=None):
        """Switch (True/False) or query (None) the folded-basis GEMM path; returns the number of folded calls so far."""
        n = _L(0)
        self._check(self._lib.gpcsd_fold_gemm(self._h, -1 if on is None else int(bool(on)), ctypes.byref(n)))
        return int(n.value)

    def ll_tridiag(self, mode=None):
        """Set (0 / False: off, 1 / True: on, 2: by size -- the default) or query (None) the shifted-tridiagonal form of the
        folded log-likelihood (no temporal eigenvectors on its path); returns the number of log-likelihoods evaluated that way."""
        n = _L(0)
        self._check(self._lib.gpcsd_ll_tridiag(self._h, -1 if mode is None else int(mode), ctypes.byref(n)))
        return int(n.value)

    def pair_share_x(self, on=None):
        """Switch (True/False) or query (None) the paired call's sharing of X = Y~ Q between its log-likelihood and its prediction
        when their temporal hyper-parameters are equal; returns the number of paired calls that shared it."""
        n = _L(0)
        self._check(self._lib.gpcsd_pair_share_x(self._h, -1 if on is None else int(bool(on)), ctypes.byref(n)))
        return int(n.value)

    def pair_share_s(self, on=None):
        """Switch (True/False) or query (None) the paired call's single spatial decomposition when its two sets differ by the jitter
        only (eigenvectors shared, spectrum shifted); returns the number of paired calls that took it."""
        n = _L(0)
        self._check(self._lib.gpcsd_pair_share_s(self._h, -1 if on is None else int(bool(on)), ctypes.byref(n)))
        if on is not None:
            self._pair_share_s_on = bool(on)
        return int(n.value)

    def pair_share_s_on(self):
        """The switch as it stands (new contexts: GPCSD_PAIR_SHARE_S, default off)."""
        return getattr(self, "_pair_share_s_on", os.environ.get("GPCSD_PAIR_SHARE_S", "0")[:1] not in ("", "0"))

    def q_pipeline(self, on=None):
        """Switch (True/False) or query (None) the panel-by-panel formation of Q and X = Y~ Q under the running tridiagonalisation
        (DESIGN 4.12); returns the number of temporal chains that took it."""
        n = _L(0)
        self._check(self._lib.gpcsd_q_pipeline(self._h, -1 if on is None else int(bool(on)), ctypes.byref(n)))
        return int(n.value)

    def bounce_stats(self):
        """Bytes moved between pageable caller memory and the device through the context's page-locked bounce blocks (the library
        never hands pageable memory to the runtime: DESIGN 6, the queue-eviction stall)."""
        n = _L(0)
        self._check(self._lib.gpcsd_bounce_stats(self._h, ctypes.byref(n)))
        return int(n.value)

    def q_pipeline_stats(self, gate_ticks=None):
        """(pipeline on?, evaluations repeated because a gate of the pipelined stage gave up -- a scheduling miss).
        gate_ticks: the gates' bound in 100 MHz ticks (0: every gate gives up at once, the test aid of the repeat path)."""
        on, n = _I(0), _L(0)
        self._check(self._lib.gpcsd_q_pipeline_stats(self._h, ctypes.byref(on), ctypes.byref(n), -1 if gate_ticks is None else int(gate_ticks)))
        return bool(on.value), int(n.value)

    def predict_chunked_copy(self, on=None):
        """Switch (True/False) or query (None) the chunked copy-out of predict()'s host arrays under its last product; returns the
        number of calls that took it."""
        n = _L(0)
        self._check(self._lib.gpcsd_predict_chunked_copy(self._h, -1 if on is None else int(bool(on)), ctypes.byref(n)))
        return int(n.value)

    def tail_early_exit(self, on=None):
        """Switch (True/False) or query (None) the rank-revealing early exit of the tridiagonalisation on positive semi-definite
        Gram matrices; returns the setting before the call."""
        prev = _I(0)
        self._check(self._lib.gpcsd_tail_early_exit(self._h, -1 if on is None else int(bool(on)), ctypes.byref(prev)))
        return bool(prev.value)

    def debug_fault_stage2(self, on):
        """Test aid: the divide & conquer stage of a staged temporal chain reports a numerical failure (late status words)."""
        self._check(self._lib.gpcsd_debug_fault_stage2(self._h, int(bool(on))))

    def decomposition_cache(self, on=None):
        """Switch (True/False) or query (None) the reuse of an unchanged side's eigendecomposition between consecutive
        fused calls; returns the number of sides reused so far."""
        n = _L(0)
        self._check(self._lib.gpcsd_decomposition_cache(self._h, -1 if on is None else int(bool(on)), ctypes.byref(n)))
        return int(n.value)

    def prof_enable(self, on=True):
        """False / 0 off; True / 1 fenced scopes; 2 asynchronous scopes (queued and paired calls stay queued, eager chains);
        3 asynchronous scopes with the chains replayed as hipGraphs (chain-level scopes only)."""
        self._check(self._lib.gpcsd_prof_enable(self._h, int(on)))

    def prof_tail_clock(self, region):
        """(ms, workgroups, flops) of the last tridiagonalisation-tail launch of a chain (0 temporal, 1 spatial, 2 other),
        from the workgroups' own wall-clock stamps; profiling modes 2 / 3, after the chain has finished."""
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._check(self._lib.gpcsd_prof_tail_clock(self._h, int(region), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
        return ms.value, n.value, fl.value

    def prof_reset(self):
        self._check(self._lib.gpcsd_prof_reset(self._h))

    def prof_names(self):
        buf = ctypes.create_string_buffer(8192)
        self._check(self._lib.gpcsd_prof_names(self._h, buf, len(buf)))
        s = buf.value.decode()
        return [n for n in s.split(";") if n]

    def prof_get(self, name):
        ms, fl = ctypes.c_double(), ctypes.c_double()
        cnt = ctypes.c_long()
        rc = self._lib.gpcsd_prof_get(self._h, name.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl))
        if rc == -2:
            return None
        self._check(rc)
        return {"ms": ms.value, "count": cnt.value, "flops": fl.value}

    def prof_all(self):
        return {n: self.prof_get(n) for n in self.prof_names()}

    def mfma_f64_peak(self):
        out = ctypes.c_double()
        self._check(self._lib.gpcsd_mfma_f64_peak(self._h, ctypes.byref(out)))
        return out.value

    def gemm_bench(self, M, N, K, transA=False, transB=False, cfg=0, reps=10):
        """(ms per launch, TFLOP/s) of the fp64 MFMA GEMM on device-resident operands."""
        out = ctypes.c_double()
        self._check(self._lib.gpcsd_gemm_bench(self._h, int(transA), int(transB), int(M), int(N), int(K), int(cfg), int(reps),
                                               ctypes.byref(out)))
        return out.value, 2.0 * M * N * K / (out.value * 1e-3) / 1e12

    def potrf_bench(self, n, reps=3):
        """(ms per factorisation, TFLOP/s in n^3 / 3 flops) of the blocked Cholesky on a device-resident SPD matrix of order n."""
        out = ctypes.c_double()
        self._check(self._lib.gpcsd_potrf_bench(self._h, int(n), int(reps), ctypes.byref(out)))
        return out.value, (n ** 3 / 3.0) / (out.value * 1e-3) / 1e12

    def potrf_diag_probe(self):
        """Phase split (us) of the 128 x 128 factor + invert workgroup of the blocked Cholesky."""
        out = np.zeros(10)
        self._check(self._lib.gpcsd_potrf_diag_probe(self._h, _ptr(out)))
        return dict(zip(["load", "serial_panels", "rank16_updates", "store_L", "diag_inverses", "level16", "level32", "level64",
                         "store_X", "total"], out.tolist()))

    def hbm_copy_peak(self, nbytes=1 << 30):
        out = ctypes.c_double()
        self._check(self._lib.gpcsd_hbm_copy_peak(self._h, int(nbytes), ctypes.byref(out)))
        return out.value


class Dist:
    """Several devices driven by ONE process through the C ABI's gpcsd_dist_* entry points (a binder without Python's
    one-process-per-GPU launcher; the class API shards through gpcsd_amd.dist.TrialSharding instead).  devices: ordinals, which
    may repeat (two contexts on one GPU)."""

    def __init__(self, devices):
        self._lib = load_library()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        rc = self._lib.gpcsd_dist_create(len(devices), devs, ctypes.byref(h))
        if rc != 0:
            raise HipUnavailable("gpcsd_dist_create failed (%d): %s" % (rc, (self._lib.gpcsd_last_error(None) or b"").decode()))
        self._h = h
        self.ndev = len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpcsd_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == 0:
            return
        msg = (self._lib.gpcsd_dist_last_error(self._h) or b"").decode()
        if rc > 0:
            raise np.linalg.LinAlgError(msg or "numerical failure (status %d)" % rc)
        if rc == ERR_CAPACITY:
            raise GPCSDCapacityError(msg)
        if rc in (-3, -22):
            raise ValueError(msg or "bad arguments")
        raise RuntimeError("libgpcsd_hip error %d: %s" % (rc, msg))

    def set_geometry_1d(self, x, gl_x, gl_w):
        x, gl_x, gl_w = _arr(x).reshape(-1), _arr(gl_x).reshape(-1), _arr(gl_w).reshape(-1)
        self._check(self._lib.gpcsd_dist_set_geometry_1d(self._h, _ptr(x), x.size, _ptr(gl_x), _ptr(gl_w), gl_x.size))

    def set_geometry_2d(self, xy, gl_x1, gl_w1, gl_x2, gl_w2):
        xy, g1, w1, g2, w2 = _arr(xy), _arr(gl_x1).reshape(-1), _arr(gl_w1).reshape(-1), _arr(gl_x2).reshape(-1), _arr(gl_w2).reshape(-1)
        self._check(self._lib.gpcsd_dist_set_geometry_2d(self._h, _ptr(xy), xy.shape[0], _ptr(g1), _ptr(w1), g1.size, _ptr(g2),
                                                         _ptr(w2), g2.size))

    def set_time(self, t):
        t = _arr(t).reshape(-1)
        self._check(self._lib.gpcsd_dist_set_time(self._h, _ptr(t), t.size))

    def set_lfp(self, lfp, replicate=False):
        lfp = _arr(lfp)
        self._check(self._lib.gpcsd_dist_set_lfp(self._h, _ptr(lfp), lfp.shape[0], lfp.shape[1], lfp.shape[2], int(bool(replicate))))
        self._shape = lfp.shape

    def loglik(self, hp):
        out = ctypes.c_double()
        self._check(self._lib.gpcsd_dist_loglik(self._h, ctypes.byref(hp), ctypes.byref(out)))
        return out.value

    def loglik_grad(self, hp, ngrad):
        out, g = np.empty(2), np.empty(ngrad)
        self._check(self._lib.gpcsd_dist_loglik_grad(self._h, ctypes.byref(hp), _ptr(out), _ptr(g), int(ngrad)))
        return float(out[0]), float(out[1]), g

    def loglik_grad_batch(self, hps, ngrad):
        B = len(hps)
        arr = (HParams * B)(*hps)
        out, g, st = np.empty((B, 2)), np.empty((B, ngrad)), np.zeros(B, dtype=np.int32)
        self._check(self._lib.gpcsd_dist_loglik_grad_batch(self._h, arr, B, _ptr(out), _ptr(g), int(ngrad),
                                                           st.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return out[:, 0].copy(), out[:, 1].copy(), g, st

    def predict(self, hp, z, tstar, type_code, want_lists=True):
        z, tstar = _arr(z), _arr(tstar).reshape(-1)
        z = z.reshape(z.shape[0], -1)
        nz, nts, R, C = z.shape[0], tstar.size, self._shape[2], hp.n_temporal
        res = {}
        ptrs = []
        for name, bit in (("csd", PRED_CSD), ("lfp", PRED_LFP)):
            if type_code & bit:
                res[name] = np.empty((nz, nts, R))
                res[name + "_list"] = np.empty((C, nz, nts, R)) if want_lists else None
            ptrs += [(_ptr(res[name + "_list"]) if res.get(name + "_list") is not None else None),
                     (_ptr(res[name]) if name in res else None)]
        self._check(self._lib.gpcsd_dist_predict(self._h, ctypes.byref(hp), _ptr(z), nz, _ptr(tstar), nts, int(type_code), *ptrs))
        return {k: v for k, v in res.items() if v is not None}


_default_ctx = None
_default_lock = threading.Lock()


def default_context():
    """Process-wide context for the stand-alone operator surface; created lazily (never at import: fork safety)."""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context()
        return _default_ctx
