"""ctypes binding of libtfhe_mi355x.so (the C ABI of include/tfhe_mi355x.h).

Used by tests/, bench.py and __graft_entry__.py; mirrors the call sequence of the reference's
server stage (`main` of server_encrypted_aes_decryption.rs:599-707): create a context, load the
evaluation keys and the transciphering key once, then transcipher batches of AES blocks.
There is no fallback: if the shared library or a HIP device is missing, calls raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
LIB_PATH = os.path.join(PKG_DIR, "libtfhe_mi355x.so")

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
f64p = C.POINTER(C.c_double)


class TxParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "n", "N", "k", "pbs_base_log", "pbs_level", "ks_base_log", "ks_level", "ks_N",
        "auto_base_log", "auto_level", "auto_split", "ss_base_log", "ss_level",
        "cbs_base_log", "cbs_level", "log_lut_count")] + [("lwe_std", C.c_double), ("glwe_std", C.c_double)]


class TxTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_keyswitch", "ms_blind_rotate", "ms_trace", "ms_scheme_switch",
                                          "ms_lut_eval", "ms_linear", "ms_total")] + \
               [(n, C.c_uint32) for n in ("n_keyswitch", "n_blind_rotate", "n_trace", "n_scheme_switch",
                                          "n_lut_eval", "n_linear", "n_br_main")] + \
               [("ms_br_main", C.c_double), ("units_br_main", C.c_uint64), ("units_blind_rotate", C.c_uint64),
                ("timing_failed", C.c_uint32), ("reserved_", C.c_uint32)]


CHUNK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t)


EXPORTS = ["tx_params_aes_tight", "tx_last_error", "tx_create", "tx_destroy", "tx_load_eval_keys", "tx_eval_key_words",
           "tx_load_round_luts", "tx_load_ctr_luts", "tx_transcipher", "tx_transcipher_partial", "tx_max_u16",
           "tx_inner_product_u16",
           "tx_stage_lwe_keyswitch", "tx_stage_blind_rotate", "tx_stage_extract_trace",
           "tx_stage_scheme_switch_fourier", "tx_stage_lut_eval", "tx_get_timing", "tx_get_twiddles",
           "tx_last_blind_rotate_kernel", "tx_ctr_shared_columns", "tx_plan_blind_rotate",
           "tx_max_u16_io", "tx_inner_product_u16_io", "tx_stage_circuit_bootstrap", "tx_timing_size",
           "tx_device_alloc", "tx_device_free", "tx_device_upload", "tx_device_download",
           "tx_transcipher_chunks", "tx_chunk_blocks", "tx_set_ctr_column_sharing", "tx_host_alloc", "tx_host_free",
           "tx_device_count", "tx_host_register", "tx_host_unregister", "tx_reserve",
           "tx_set_decomposer_tie_rule"]

# the experiment build: the same sources with -DTXD_EXPERIMENTS (superseded kernels + TX_*_VARIANT knobs); tests and A/B tools only
EXP_LIB_PATH = os.path.join(PKG_DIR, "libtfhe_mi355x_exp.so")
_LIB = None
_LIBS = {}


class TxError(RuntimeError):
    pass


def lib(experiments=False):
    global _LIB
    path = EXP_LIB_PATH if experiments else LIB_PATH
    if path not in _LIBS:
        if not os.path.exists(path):
            raise TxError("%s is not built (run __graft_entry__.build()); no fallback exists" % os.path.basename(path))
        L = C.CDLL(path)
        L.tx_last_error.restype = C.c_char_p
        L.tx_create.restype = C.c_void_p
        L.tx_create.argtypes = [C.POINTER(TxParams), C.c_int]
        L.tx_destroy.argtypes = [C.c_void_p]
        L.tx_load_eval_keys.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t]
        L.tx_eval_key_words.argtypes = [C.c_void_p, C.POINTER(C.c_size_t * 4)]
        L.tx_load_round_luts.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.tx_load_ctr_luts.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.tx_transcipher.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, C.c_void_p, C.c_int]
        L.tx_transcipher_partial.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_int, u64p]
        L.tx_max_u16.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
        L.tx_inner_product_u16.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
        L.tx_max_u16_io.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.tx_inner_product_u16_io.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.tx_stage_circuit_bootstrap.argtypes = [C.c_void_p, u64p, C.c_size_t, f64p]
        L.tx_device_alloc.restype = C.c_void_p
        L.tx_device_alloc.argtypes = [C.c_void_p, C.c_size_t]
        L.tx_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.tx_device_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.tx_device_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.tx_timing_size.restype = C.c_size_t
        L.tx_timing_size.argtypes = []
        if L.tx_timing_size() != C.sizeof(TxTiming):
            raise TxError("libtfhe_mi355x.so and txhip.py disagree about tx_timing (%d vs %d bytes): rebuild"
                          % (L.tx_timing_size(), C.sizeof(TxTiming)))
        L.tx_stage_lwe_keyswitch.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
        L.tx_stage_blind_rotate.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
        L.tx_stage_extract_trace.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
        L.tx_stage_scheme_switch_fourier.argtypes = [C.c_void_p, u64p, C.c_size_t, f64p]
        L.tx_stage_lut_eval.argtypes = [C.c_void_p, f64p, C.c_size_t, C.c_int, C.c_int, C.c_int, u64p]
        L.tx_get_timing.argtypes = [C.c_void_p, C.POINTER(TxTiming)]
        L.tx_get_twiddles.argtypes = [C.c_void_p, f64p]
        L.tx_ctr_shared_columns.argtypes = [u8p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.tx_plan_blind_rotate.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        L.tx_last_blind_rotate_kernel.restype = C.c_char_p
        L.tx_last_blind_rotate_kernel.argtypes = [C.c_void_p]
        L.tx_transcipher_chunks.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, C.c_void_p, C.c_int, CHUNK_FN, C.c_void_p]
        L.tx_chunk_blocks.restype = C.c_size_t
        L.tx_chunk_blocks.argtypes = []
        L.tx_set_ctr_column_sharing.argtypes = [C.c_void_p, C.c_int]
        L.tx_host_alloc.restype = C.c_void_p
        L.tx_host_alloc.argtypes = [C.c_size_t]
        L.tx_host_free.argtypes = [C.c_void_p]
        L.tx_device_count.argtypes = []
        L.tx_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.tx_host_unregister.argtypes = [C.c_void_p]
        L.tx_reserve.argtypes = [C.c_void_p, C.c_size_t]
        L.tx_set_decomposer_tie_rule.argtypes = [C.c_void_p, C.c_int]
        _LIBS[path] = L
        if not experiments:
            _LIB = L
    return _LIBS[path]


def _p64(a): return a.ctypes.data_as(u64p)
def _p8(a): return a.ctypes.data_as(u8p)
def _pf(a): return a.ctypes.data_as(f64p)


def aes_tight(n=768):
    p = TxParams()
    lib().tx_params_aes_tight(C.byref(p))
    p.n = n
    return p


def plan_blind_rotate(count, num_cus=256):
    """[(kernel number, units), ...]: the launches run_br splits a blind rotation of `count` units into (host logic, no GPU)"""
    v = (C.c_int * 2)(); n = (C.c_size_t * 2)()
    k = lib().tx_plan_blind_rotate(count, num_cus, v, n)
    if k < 0:
        raise TxError(lib().tx_last_error().decode())
    return [(int(v[i]), int(n[i])) for i in range(k)]


def ctr_shared_columns(iv, first_block, n_blocks):
    """(rep_of [n_blocks * 4], col_of [distinct]) of the CTR schedule's second round (host logic, no GPU)"""
    ivb = np.frombuffer(bytes(iv), np.uint8).copy()
    rep = np.zeros(n_blocks * 4, np.int32)
    col = np.zeros(n_blocks * 4, np.int32)
    k = lib().tx_ctr_shared_columns(_p8(ivb), first_block, n_blocks, rep.ctypes.data_as(C.POINTER(C.c_int)), col.ctypes.data_as(C.POINTER(C.c_int)))
    if k < 0:
        raise TxError(lib().tx_last_error().decode())
    return rep, col[:k]


class Engine:
    """One context on one GPU (tx_ctx)."""

    def __init__(self, params=None, device=0, experiments=False):
        """experiments=True: the -DTXD_EXPERIMENTS build (libtfhe_mi355x_exp.so), which also holds the superseded kernels and
        reads the TX_*_VARIANT knobs from the environment; the product library reads none"""
        L = self.L = lib(experiments)
        self.h = None
        self.p = params or aes_tight()
        self.h = L.tx_create(C.byref(self.p), device)
        if not self.h:
            raise TxError(L.tx_last_error().decode())
        self.K = self.p.k + 1
        self.gl = self.K * self.p.N
        self.big = self.p.k * self.p.N + 1
        self.small = self.p.n + 1

    def close(self):
        if self.h:
            self.L.tx_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise TxError(self.L.tx_last_error().decode())

    def eval_key_words(self):
        """word counts (bsk, ksk, auto, ss) the context expects"""
        out = (C.c_size_t * 4)()
        self._chk(self.L.tx_eval_key_words(self.h, C.byref(out)))
        return tuple(out)

    def load_eval_keys(self, bsk, ksk, auto_keys, ss_key):
        ks = [np.ascontiguousarray(k) for k in (bsk, ksk, auto_keys, ss_key)]
        for k, name in zip(ks, ("bsk", "ksk", "auto_keys", "ss_key")):
            if k.dtype != np.uint64:
                raise TxError("load_eval_keys: %s must be uint64 torus words, got %s" % (name, k.dtype))
        # the library checks every length against the context's parameters before it reads anything
        self._chk(self.L.tx_load_eval_keys(self.h, _p64(ks[0]), ks[0].size, _p64(ks[1]), ks[1].size,
                                          _p64(ks[2]), ks[2].size, _p64(ks[3]), ks[3].size))

    def load_round_luts(self, luts):
        self._chk(self.L.tx_load_round_luts(self.h, _p64(luts), luts.size))

    def load_ctr_luts(self, luts):
        self._chk(self.L.tx_load_ctr_luts(self.h, _p64(luts), luts.size))

    def transcipher(self, ct_bytes, iv=None, out_device_ptr=None):
        ct = np.frombuffer(bytes(ct_bytes), np.uint8).copy()
        nb = ct.size // 16
        ivp = None
        if iv is not None:
            ivb = np.frombuffer(bytes(iv), np.uint8).copy()
            ivp = _p8(ivb)
        if out_device_ptr is not None:
            self._chk(self.L.tx_transcipher(self.h, _p8(ct), nb, ivp, C.c_void_p(out_device_ptr), 1))
            return None
        out = np.zeros(nb * 128 * self.big, np.uint64)
        self._chk(self.L.tx_transcipher(self.h, _p8(ct), nb, ivp, out.ctypes.data_as(C.c_void_p), 0))
        return out.reshape(nb * 128, self.big)

    def transcipher_chunks(self, ct_bytes, iv=None, pinned=True):
        """tx_transcipher_chunks with a host result: (LWEs [nb * 128, big], [(first_block, n_blocks), ...] as the callback saw
        them, and for each callback whether the chunk's rows were complete in the result at that moment)"""
        ct = np.frombuffer(bytes(ct_bytes), np.uint8).copy()
        nb = ct.size // 16
        ivp = None
        if iv is not None:
            ivb = np.frombuffer(bytes(iv), np.uint8).copy()
            ivp = _p8(ivb)
        nbytes = nb * 128 * self.big * 8
        hp, reg = None, None
        if pinned == "register":          # the caller's own memory, page-locked in place (tx_host_register)
            out = np.empty(nb * 128 * self.big, np.uint64)
            self._chk(self.L.tx_host_register(out.ctypes.data_as(C.c_void_p), nbytes))
            reg = out.ctypes.data
        elif pinned:
            hp = self.L.tx_host_alloc(nbytes)
            if not hp:
                raise TxError(self.L.tx_last_error().decode())
            out = np.ctypeslib.as_array(C.cast(hp, u64p), shape=(nb * 128 * self.big,))
        else:
            out = np.empty(nb * 128 * self.big, np.uint64)
        out[:] = 0xDEADBEEFDEADBEEF
        seen, arrived = [], []

        def on_chunk(user, first, count):
            seen.append((int(first), int(count)))
            rows = out[first * 128 * self.big:(first + count) * 128 * self.big].reshape(count * 128, self.big)
            arrived.append(bool((rows[:, -1] != 0xDEADBEEFDEADBEEF).all() and (rows[:, 0] != 0xDEADBEEFDEADBEEF).all()))
        cb = CHUNK_FN(on_chunk)
        try:
            self._chk(self.L.tx_transcipher_chunks(self.h, _p8(ct), nb, ivp, out.ctypes.data_as(C.c_void_p), 0, cb, None))
            res = out.reshape(nb * 128, self.big).copy()
        finally:
            if reg:
                self.L.tx_host_unregister(C.c_void_p(reg))
            if hp:
                del out
                self.L.tx_host_free(C.c_void_p(hp))
        return res, seen, arrived

    def reserve(self, n_blocks):
        self._chk(self.L.tx_reserve(self.h, n_blocks))

    def set_decomposer_tie_rule(self, tfhe_rs):
        """True: every decomposition follows tfhe-rs' SignedDecomposer (top-level tie always +B/2); False: balanced (default)"""
        self._chk(self.L.tx_set_decomposer_tie_rule(self.h, 1 if tfhe_rs else 0))

    def set_ctr_column_sharing(self, enabled):
        self._chk(self.L.tx_set_ctr_column_sharing(self.h, 1 if enabled else 0))

    def transcipher_partial(self, ct_bytes, rounds_done):
        ct = np.frombuffer(bytes(ct_bytes), np.uint8).copy()
        nb = ct.size // 16
        out = np.zeros(nb * 128 * self.big, np.uint64)
        self._chk(self.L.tx_transcipher_partial(self.h, _p8(ct), nb, rounds_done, _p64(out)))
        return out.reshape(nb * 128, self.big)

    def max_u16(self, lwe_bits):
        lwe_bits = np.ascontiguousarray(lwe_bits, np.uint64)
        nv = lwe_bits.size // (16 * self.big)
        out = np.zeros(16 * self.big, np.uint64)
        self._chk(self.L.tx_max_u16(self.h, _p64(lwe_bits), nv, _p64(out)))
        return out.reshape(16, self.big)

    def inner_product_u16(self, lwe_bits):
        lwe_bits = np.ascontiguousarray(lwe_bits, np.uint64)
        nv = lwe_bits.size // (16 * self.big)
        out = np.zeros(16 * self.big, np.uint64)
        self._chk(self.L.tx_inner_product_u16(self.h, _p64(lwe_bits), nv, _p64(out)))
        return out.reshape(16, self.big)

    # ---- device memory of this context's GPU (no torch needed) ----
    def device_alloc(self, nbytes):
        p = self.L.tx_device_alloc(self.h, nbytes)
        if not p:
            raise TxError(self.L.tx_last_error().decode())
        return p

    def device_free(self, ptr):
        self._chk(self.L.tx_device_free(self.h, C.c_void_p(ptr)))

    def device_upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.tx_device_upload(self.h, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def device_download(self, ptr, shape, dtype=np.uint64):
        out = np.zeros(shape, dtype)
        self._chk(self.L.tx_device_download(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    def max_u16_device(self, in_device_ptr, n_values, out_device_ptr):
        """max of n_values u16 whose LWE bits are device memory on this context's GPU; the 16 result LWEs go to device memory"""
        self._chk(self.L.tx_max_u16_io(self.h, C.c_void_p(in_device_ptr), n_values, C.c_void_p(out_device_ptr), 1, 1))

    def inner_product_u16_device(self, in_device_ptr, n_values, out_device_ptr):
        self._chk(self.L.tx_inner_product_u16_io(self.h, C.c_void_p(in_device_ptr), n_values, C.c_void_p(out_device_ptr), 1, 1))

    def max_u16_io(self, lwe_bits_or_ptr, n_values, in_is_device, out_device_ptr=None):
        """any combination of host / device input and output (tests): a host input is a uint64 array, a host output is returned"""
        if in_is_device:
            inp = C.c_void_p(lwe_bits_or_ptr)
        else:
            arr = np.ascontiguousarray(lwe_bits_or_ptr, np.uint64)
            inp = arr.ctypes.data_as(C.c_void_p)
        if out_device_ptr is not None:
            self._chk(self.L.tx_max_u16_io(self.h, inp, n_values, C.c_void_p(out_device_ptr), int(in_is_device), 1))
            return None
        out = np.zeros(16 * self.big, np.uint64)
        self._chk(self.L.tx_max_u16_io(self.h, inp, n_values, out.ctypes.data_as(C.c_void_p), int(in_is_device), 0))
        return out.reshape(16, self.big)

    def inner_product_u16_io(self, lwe_bits_or_ptr, n_values, in_is_device, out_device_ptr=None):
        if in_is_device:
            inp = C.c_void_p(lwe_bits_or_ptr)
        else:
            arr = np.ascontiguousarray(lwe_bits_or_ptr, np.uint64)
            inp = arr.ctypes.data_as(C.c_void_p)
        if out_device_ptr is not None:
            self._chk(self.L.tx_inner_product_u16_io(self.h, inp, n_values, C.c_void_p(out_device_ptr), int(in_is_device), 1))
            return None
        out = np.zeros(16 * self.big, np.uint64)
        self._chk(self.L.tx_inner_product_u16_io(self.h, inp, n_values, out.ctypes.data_as(C.c_void_p), int(in_is_device), 0))
        return out.reshape(16, self.big)

    # ---- stage entry points ----
    def circuit_bootstrap(self, lwe_big):
        """a4-a9 as the pipeline runs them (fused trace + scheme switch kernel): Fourier GGSWs [count, level, K, K, N]"""
        lwe_big = np.ascontiguousarray(lwe_big, np.uint64)
        cnt = lwe_big.size // self.big
        out = np.zeros(cnt * self.p.cbs_level * self.K * self.gl, np.float64)
        self._chk(self.L.tx_stage_circuit_bootstrap(self.h, _p64(lwe_big), cnt, _pf(out)))
        return out.reshape(cnt, self.p.cbs_level, self.K, self.K, self.p.N)

    def lwe_keyswitch(self, lwe_big):
        lwe_big = np.ascontiguousarray(lwe_big, np.uint64)
        cnt = lwe_big.size // self.big
        out = np.zeros(cnt * self.small, np.uint64)
        self._chk(self.L.tx_stage_lwe_keyswitch(self.h, _p64(lwe_big), cnt, _p64(out)))
        return out.reshape(cnt, self.small)

    def blind_rotate(self, lwe_small):
        lwe_small = np.ascontiguousarray(lwe_small, np.uint64)
        cnt = lwe_small.size // self.small
        out = np.zeros(cnt * self.gl, np.uint64)
        self._chk(self.L.tx_stage_blind_rotate(self.h, _p64(lwe_small), cnt, _p64(out)))
        return out.reshape(cnt, self.K, self.p.N)

    def extract_trace(self, acc):
        acc = np.ascontiguousarray(acc, np.uint64)
        cnt = acc.size // self.gl
        out = np.zeros(cnt * self.p.cbs_level * self.gl, np.uint64)
        self._chk(self.L.tx_stage_extract_trace(self.h, _p64(acc), cnt, _p64(out)))
        return out.reshape(cnt, self.p.cbs_level, self.K, self.p.N)

    def scheme_switch_fourier(self, glev):
        glev = np.ascontiguousarray(glev, np.uint64)
        cnt = glev.size // (self.p.cbs_level * self.gl)
        out = np.zeros(cnt * self.p.cbs_level * self.K * self.gl, np.float64)
        self._chk(self.L.tx_stage_scheme_switch_fourier(self.h, _p64(glev), cnt, _pf(out)))
        return out.reshape(cnt, self.p.cbs_level, self.K, self.K, self.p.N)

    def lut_eval(self, ggsw_f, ctr, lut_set, nlut):
        ggsw_f = np.ascontiguousarray(ggsw_f, np.float64)
        nbytes = ggsw_f.size // (8 * self.p.cbs_level * self.K * self.gl)
        out = np.zeros(nlut * nbytes * 8 * self.big, np.uint64)
        self._chk(self.L.tx_stage_lut_eval(self.h, _pf(ggsw_f), nbytes, int(ctr), lut_set, nlut, _p64(out)))
        return out.reshape(nlut, nbytes * 8, self.big)

    def timing(self):
        t = TxTiming()
        self._chk(self.L.tx_get_timing(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in TxTiming._fields_}

    def last_blind_rotate_kernel(self):
        return self.L.tx_last_blind_rotate_kernel(self.h).decode()

    def twiddles(self):
        out = np.zeros(15 * 64 * 2, np.float64)
        self._chk(self.L.tx_get_twiddles(self.h, _pf(out)))
        return out
