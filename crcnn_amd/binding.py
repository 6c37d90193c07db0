"""ctypes binding of libcrcnn_hip.so (the C ABI in include/crcnn_hip.h).

This module is plumbing: it loads the in-tree shared library, checks that every symbol the header declares is
exported, and offers a small `Engine` convenience class (device buffers + numpy round trips) for tests, the bench
harness and Python users.  There is NO CPU fallback: a missing library or a failing call raises.
"""
import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcrcnn_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "crcnn_hip.h")

u64 = ctypes.c_uint64
PU = ctypes.POINTER(u64)
VP = ctypes.c_void_p
CI = ctypes.c_int
SZ = ctypes.c_size_t

COEFF, NTT, NTTP, NTTL, NTTL1, NTTLC = 0, 1, 2, 3, 4, 5


class CrcError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        super().__init__(f"{what}: {_strerror(status)} (status {status}, hip error {_lib.crc_last_hip_error() if _lib else '?'})")


_lib = None


def header_symbols():
    """names of every function declared in include/crcnn_hip.h"""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(crc_[a-z0-9_]+)\s*\(", txt)))


def _strerror(s):
    return _lib.crc_strerror(s).decode() if _lib else "?"


def load():
    """dlopen the in-tree library; raises if it was not built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built (no CPU fallback exists)")
    try:
        # torch ships its own libamdhip64: load it first so that this library binds to the same HIP runtime (two runtimes in one
        # process cannot both own the device: whichever initialises second sees "no HIP GPUs")
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(L, s)]
    if missing:
        raise ImportError(f"libcrcnn_hip.so does not export: {missing}")
    L.crc_strerror.restype = ctypes.c_char_p
    L.crc_strerror.argtypes = [CI]
    L.crc_ctx_create.argtypes = [CI, PU, CI, u64, CI, ctypes.POINTER(VP)]
    L.crc_ctx_destroy.argtypes = [VP]
    L.crc_ctx_destroy.restype = None
    L.crc_default_coeff_modulus_128.argtypes = [CI, PU, CI]
    for f in ("crc_ctx_n", "crc_ctx_k", "crc_ctx_kbsk", "crc_ctx_device"):
        getattr(L, f).argtypes = [VP]
    L.crc_ct_words.restype = SZ; L.crc_ct_words.argtypes = [VP, CI]
    L.crc_evk_words.restype = SZ; L.crc_evk_words.argtypes = [VP, CI]
    L.crc_ctx_table.argtypes = [VP, ctypes.c_char_p, PU, CI]
    L.crc_ctx_set_tuning.argtypes = [VP, ctypes.c_char_p, ctypes.c_longlong]
    L.crc_mem_info.argtypes = [VP, ctypes.POINTER(SZ), ctypes.POINTER(SZ)]
    L.crc_malloc.argtypes = [VP, SZ, ctypes.POINTER(VP)]
    L.crc_free.argtypes = [VP, VP]
    L.crc_memcpy_h2d.argtypes = [VP, VP, VP, SZ, VP]
    L.crc_memcpy_d2h.argtypes = [VP, VP, VP, SZ, VP]
    L.crc_memcpy_d2d.argtypes = [VP, VP, VP, SZ, VP]
    L.crc_memset.argtypes = [VP, VP, CI, SZ, VP]
    L.crc_stream_sync.argtypes = [VP, VP]
    L.crc_event_create.argtypes = [VP, ctypes.POINTER(VP)]; L.crc_event_destroy.argtypes = [VP, VP]; L.crc_event_record.argtypes = [VP, VP, VP]
    L.crc_event_elapsed_ms.argtypes = [VP, VP, VP, ctypes.POINTER(ctypes.c_float)]
    L.crc_stream_create.argtypes = [VP, ctypes.POINTER(VP)]; L.crc_stream_destroy.argtypes = [VP, VP]; L.crc_stream_wait_event.argtypes = [VP, VP, VP]
    L.crc_host_alloc.argtypes = [VP, SZ, ctypes.POINTER(VP)]; L.crc_host_free.argtypes = [VP, VP]
    L.crc_host_thread_limit.argtypes = []
    L.crc_encode_f32.argtypes = [VP, ctypes.POINTER(ctypes.c_float), SZ, PU, ctypes.POINTER(ctypes.c_int32)]
    L.crc_encode_f64.argtypes = [VP, ctypes.POINTER(ctypes.c_double), SZ, PU, ctypes.POINTER(ctypes.c_int32)]
    L.crc_decode.restype = ctypes.c_double; L.crc_decode.argtypes = [VP, PU]
    L.crc_bn_invstd_f32.argtypes = [ctypes.POINTER(ctypes.c_float), SZ, ctypes.POINTER(ctypes.c_float)]
    L.crc_plain_to_ntt.argtypes = [VP, VP, SZ, VP, VP]
    L.crc_encode_f32_compact.argtypes = [VP, ctypes.POINTER(ctypes.c_float), SZ, PU, ctypes.POINTER(ctypes.c_int32)]
    L.crc_plain_expand.argtypes = [VP, VP, SZ, VP, VP]
    L.crc_plain_to_delta.argtypes = [VP, VP, SZ, CI, VP, VP]
    L.crc_ntt_fwd.argtypes = [VP, VP, SZ, CI, VP]
    L.crc_ntt_inv.argtypes = [VP, VP, SZ, CI, VP]
    L.crc_ntt_fwd_bsk.argtypes = [VP, VP, SZ, VP]
    L.crc_ntt_inv_bsk.argtypes = [VP, VP, SZ, VP]
    L.crc_add.argtypes = [VP, VP, VP, SZ, CI, VP]
    L.crc_add_plain.argtypes = [VP, VP, VP, SZ, SZ, CI, VP]
    L.crc_multiply_plain_ntt.argtypes = [VP, VP, VP, SZ, SZ, CI, VP]
    L.crc_multiply_plain.argtypes = [VP, VP, VP, SZ, SZ, VP]
    L.crc_conv2d_work_bytes.restype = SZ; L.crc_conv2d_work_bytes.argtypes = [VP] + [CI] * 10
    L.crc_conv2d.argtypes = [VP, VP, VP, VP] + [CI] * 11 + [VP, VP, VP]
    L.crc_conv2d_fold_pool.argtypes = [VP, VP, VP, VP] + [CI] * 8 + [VP, VP, VP]
    L.crc_dense_work_bytes.restype = SZ; L.crc_dense_work_bytes.argtypes = [VP, CI, CI, CI, CI]
    L.crc_dense.argtypes = [VP, VP, VP, VP, CI, CI, CI, CI, CI, VP, VP, VP]
    L.crc_pool.argtypes = [VP, VP] + [CI] * 8 + [VP, CI, VP, VP]
    L.crc_batchnorm.argtypes = [VP, VP, CI, CI, CI, CI, VP, VP, CI, VP]
    L.crc_square_relin_work_bytes.restype = SZ; L.crc_square_relin_work_bytes.argtypes = [VP, SZ, CI]
    L.crc_square_relin.argtypes = [VP, VP, SZ, VP, CI, VP, VP, VP]
    L.crc_square_relin_forms.argtypes = [VP, VP, CI, SZ, VP, CI, VP, CI, VP, VP]
    L.crc_square_pool_relin_supported.argtypes = [VP, CI, CI, CI]
    L.crc_square_pool_relin_work_bytes.restype = SZ; L.crc_square_pool_relin_work_bytes.argtypes = [VP] + [CI] * 9
    L.crc_square_pool_relin_forms.argtypes = [VP, VP, CI] + [CI] * 8 + [VP, CI, VP, VP, CI, VP, VP]
    L.crc_conv2d_forms.argtypes = [VP, VP, VP, CI, VP] + [CI] * 9 + [CI, CI, VP, VP, VP]
    L.crc_dense_forms.argtypes = [VP, VP, VP, CI, VP, CI, CI, CI, CI, CI, VP, VP, VP]
    L.crc_pack28.argtypes = [VP, VP, SZ, CI, VP]
    L.crc_limb_supported.argtypes = [VP, CI, CI, CI]
    L.crc_limb_tensor_bytes.restype = SZ; L.crc_limb_tensor_bytes.argtypes = [VP, CI, CI, CI, CI]
    L.crc_limb_weights_bytes.restype = SZ; L.crc_limb_weights_bytes.argtypes = [VP, CI, CI, CI, CI]
    L.crc_limb_pack_weights.argtypes = [VP, VP, CI, CI, CI, CI, VP, VP]
    L.crc_limb_pack_weights_tile.argtypes = [VP, VP, CI, CI, CI, CI, CI, CI, VP, VP]
    L.crc_limb_pack_tensor_at.argtypes = [VP, VP, CI, CI, CI, CI, CI, VP, CI, CI, VP]
    L.crc_plan_mac.argtypes = [VP] + [CI] * 10 + [ctypes.POINTER(CI)]
    L.crc_plan_fold_pool.argtypes = [VP] + [CI] * 12 + [ctypes.POINTER(CI)]
    L.crc_limb_pack_tensor.argtypes = [VP, VP, CI, CI, CI, CI, CI, VP, VP]
    L.crc_limb_conv1_supported.argtypes = [VP] + [CI] * 8
    L.crc_limb_conv1_weights_bytes.restype = SZ; L.crc_limb_conv1_weights_bytes.argtypes = [VP]
    L.crc_limb_conv1_pack_weights.argtypes = [VP, VP, CI, CI, CI, VP, VP]
    L.crc_conv2d_forms_work_bytes.restype = SZ; L.crc_conv2d_forms_work_bytes.argtypes = [VP] + [CI] * 12
    L.crc_square.argtypes = [VP, VP, SZ, VP, VP, VP]
    L.crc_relinearize.argtypes = [VP, VP, SZ, VP, CI, VP, VP, VP]
    L.crc_import_seal.argtypes = [VP, PU, CI, PU]
    L.crc_export_seal.argtypes = [VP, PU, CI, PU]
    L.crc_params_hash.argtypes = [VP, PU]
    for f_ in ("ct", "evk", "pk", "sk"):
        getattr(L, f"crc_seal_{f_}_bytes").restype = SZ
    L.crc_seal_ct_bytes.argtypes = [VP, CI]; L.crc_seal_evk_bytes.argtypes = [VP, CI]; L.crc_seal_pk_bytes.argtypes = [VP]; L.crc_seal_sk_bytes.argtypes = [VP]
    L.crc_seal_ct_save.argtypes = [VP, PU, CI, VP, SZ, ctypes.POINTER(SZ)]
    L.crc_seal_ct_load.argtypes = [VP, VP, SZ, PU, CI, ctypes.POINTER(CI), ctypes.POINTER(SZ)]
    L.crc_seal_evk_save.argtypes = [VP, PU, CI, VP, SZ, ctypes.POINTER(SZ)]
    L.crc_seal_evk_load.argtypes = [VP, VP, SZ, PU, ctypes.POINTER(CI)]
    L.crc_seal_pk_save.argtypes = [VP, PU, VP, SZ, ctypes.POINTER(SZ)]
    L.crc_seal_pk_load.argtypes = [VP, VP, SZ, PU]
    L.crc_seal_sk_save.argtypes = [VP, PU, VP, SZ, ctypes.POINTER(SZ)]
    L.crc_seal_sk_load.argtypes = [VP, VP, SZ, PU]
    L.crc_h5_dataset_count.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(SZ)]
    L.crc_h5_read_f32.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), SZ, ctypes.POINTER(SZ)]
    L.crc_h5_list.argtypes = [ctypes.c_char_p, ctypes.c_char_p, SZ]
    L.crc_keygen.argtypes = [VP, u64, PU, PU]
    L.crc_gen_evk.argtypes = [VP, u64, PU, CI, PU]
    L.crc_encrypt.argtypes = [VP, PU, PU, SZ, u64, PU]
    L.crc_encrypt_dev_work_bytes.restype = SZ; L.crc_encrypt_dev_work_bytes.argtypes = [VP, SZ]
    L.crc_encrypt_dev.argtypes = [VP, VP, VP, SZ, u64, VP, VP, VP]
    PB = ctypes.POINTER(ctypes.c_uint8)
    L.crc_random_key.argtypes = [PB]
    L.crc_chacha20_block.argtypes = [PB, ctypes.c_uint32, PB, PB]
    L.crc_keygen_key.argtypes = [VP, PB, PU, PU]
    L.crc_gen_evk_key.argtypes = [VP, PB, PU, CI, PU]
    L.crc_encrypt_key.argtypes = [VP, PU, PU, SZ, PB, u64, PU]
    L.crc_encrypt_dev_key.argtypes = [VP, VP, VP, SZ, PB, u64, VP, VP, VP]
    L.crc_encrypt_dev_forms.argtypes = [VP, VP, VP, SZ, u64, ctypes.c_int, VP, VP, VP]
    L.crc_encrypt_dev_key_forms.argtypes = [VP, VP, VP, SZ, PB, u64, ctypes.c_int, VP, VP, VP]
    L.crc_encrypt_dev_noise_thresholds.restype = None; L.crc_encrypt_dev_noise_thresholds.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.crc_decrypt_dev_work_bytes.restype = SZ; L.crc_decrypt_dev_work_bytes.argtypes = [VP, SZ, CI, CI]
    L.crc_decrypt_dev.argtypes = [VP, VP, VP, SZ, CI, CI, VP, VP, VP]
    L.crc_decode_dev.argtypes = [VP, VP, SZ, VP, VP]
    L.crc_encode_dev_f32.argtypes = [VP, VP, SZ, VP, VP]
    L.crc_encode_dev_f64.argtypes = [VP, VP, SZ, VP, VP]
    L.crc_refresh_dev_work_bytes.restype = SZ; L.crc_refresh_dev_work_bytes.argtypes = [VP, SZ, CI]
    L.crc_refresh_dev.argtypes = [VP, VP, VP, VP, SZ, CI, u64, CI, VP, VP, VP, VP]
    L.crc_refresh_dev_key.argtypes = [VP, VP, VP, VP, SZ, CI, PB, u64, CI, VP, VP, VP, VP]
    L.crc_comm_unique_id.argtypes = [PB]
    L.crc_comm_create.argtypes = [VP, CI, CI, PB, ctypes.POINTER(VP)]
    L.crc_comm_create_all.argtypes = [ctypes.POINTER(VP), CI, ctypes.POINTER(VP)]
    L.crc_comm_destroy.argtypes = [VP]; L.crc_comm_destroy.restype = None
    L.crc_comm_rank.argtypes = [VP]; L.crc_comm_world.argtypes = [VP]
    L.crc_broadcast_weights.argtypes = [VP, VP, SZ, CI, VP]
    L.crc_broadcast_weights_all.argtypes = [ctypes.POINTER(VP), CI, ctypes.POINTER(VP), SZ, CI, ctypes.POINTER(VP)]
    L.crc_comm_allgather_u64.argtypes = [VP, PU, SZ, PU, VP]
    L.crc_checksum64.argtypes = [VP, VP, SZ, PU, VP]
    L.crc_decrypt.argtypes = [VP, PU, PU, SZ, CI, PU]
    L.crc_noise_budget.argtypes = [VP, PU, PU, CI]
    _lib = L
    return L


def _chk(status, what):
    if status < 0:
        raise CrcError(status, what)
    return status


def _pu(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(PU)


def default_coeff_modulus_128(n):
    L = load()
    buf = (u64 * 16)()
    cnt = _chk(L.crc_default_coeff_modulus_128(n, buf, 16), "crc_default_coeff_modulus_128")
    return [int(buf[i]) for i in range(cnt)]


def h5_read(path, name):
    L = load()
    cnt = SZ(0)
    _chk(L.crc_h5_dataset_count(path.encode(), name.encode(), ctypes.byref(cnt)), f"crc_h5_dataset_count({name})")
    out = np.zeros(cnt.value, dtype=np.float32)
    _chk(L.crc_h5_read_f32(path.encode(), name.encode(), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), cnt.value, None), "crc_h5_read_f32")
    return out


def h5_list(path):
    L = load()
    buf = ctypes.create_string_buffer(1 << 16)
    _chk(L.crc_h5_list(path.encode(), buf, len(buf)), "crc_h5_list")
    return [s for s in buf.value.decode().split("\n") if s]


class DBuf:
    """a device allocation owned through crc_malloc/crc_free"""

    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, int(nbytes)
        p = VP()
        _chk(eng.L.crc_malloc(eng.c, max(self.nbytes, 8), ctypes.byref(p)), "crc_malloc")
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.eng.L.crc_free(self.eng.c, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One context = one (n, q[], t) parameter set on one device (device=-1: host-only, for encode/client/tables)."""

    def __init__(self, n, q, t, device=0):
        self.L = load()
        self.n, self.k, self.t, self.device = int(n), len(q), int(t), device
        self.q = np.array(q, dtype=np.uint64)
        c = VP()
        _chk(self.L.crc_ctx_create(self.n, _pu(self.q), self.k, self.t, device, ctypes.byref(c)), "crc_ctx_create")
        self.c = c
        self.kbsk = self.L.crc_ctx_kbsk(self.c)
        self.stream = None

    def close(self):
        if self.c:
            self.L.crc_ctx_destroy(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory
    def mem_info(self):
        f, t = SZ(0), SZ(0)
        _chk(self.L.crc_mem_info(self.c, ctypes.byref(f), ctypes.byref(t)), "crc_mem_info")
        return f.value, t.value

    def alloc(self, nbytes):
        return DBuf(self, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        b = DBuf(self, arr.nbytes)
        _chk(self.L.crc_memcpy_h2d(self.c, b.ptr, arr.ctypes.data, arr.nbytes, self.stream), "crc_memcpy_h2d")
        self.sync()
        return b

    def download(self, buf, shape, dtype=np.uint64):
        out = np.zeros(shape, dtype=dtype)
        self.sync()
        _chk(self.L.crc_memcpy_d2h(self.c, out.ctypes.data, buf.ptr if isinstance(buf, DBuf) else buf, out.nbytes, self.stream), "crc_memcpy_d2h")
        self.sync()
        return out

    def sync(self):
        _chk(self.L.crc_stream_sync(self.c, self.stream), "crc_stream_sync")

    def set_tuning(self, name, value):
        """tools / tests only: change one tuning switch of this (quiescent) context (the environment is read once, at creation)"""
        self.sync()
        _chk(self.L.crc_ctx_set_tuning(self.c, name.encode(), int(value)), f"crc_ctx_set_tuning({name})")

    def table(self, name, cap=1 << 16):
        out = np.zeros(cap, dtype=np.uint64)
        cnt = _chk(self.L.crc_ctx_table(self.c, name.encode(), _pu(out), cap), f"crc_ctx_table({name})")
        return out[:cnt].copy()

    # ---- host side: encode / client
    def encode(self, values, dtype=np.float32):
        values = np.ascontiguousarray(np.asarray(values, dtype=dtype).reshape(-1))
        out = np.zeros((values.size, self.n), dtype=np.uint64)
        cc = np.zeros(values.size, dtype=np.int32)
        if dtype == np.float32:
            _chk(self.L.crc_encode_f32(self.c, values.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), values.size, _pu(out), cc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))), "crc_encode_f32")
        else:
            _chk(self.L.crc_encode_f64(self.c, values.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), values.size, _pu(out), cc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))), "crc_encode_f64")
        return out, cc

    COMPACT_WORDS = 96        # CRC_PLAIN_COMPACT_WORDS

    def encode_to_device(self, values, d_plain, d_compact):
        """float32 weights -> dense coefficient-form plaintexts [count][n] at d_plain: encoded in compact form on the host threads (96 words per weight, the only
        coefficients the encoder sets), copied down as they are and zero-extended on the device.  d_compact: count * 96 * 8 bytes of device staging"""
        values = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
        cp = np.empty((values.size, self.COMPACT_WORDS), dtype=np.uint64)
        _chk(self.L.crc_encode_f32_compact(self.c, values.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), values.size, _pu(cp), None), "crc_encode_f32_compact")
        _chk(self.L.crc_memcpy_h2d(self.c, self.p(d_compact), cp.ctypes.data, cp.nbytes, self.stream), "crc_memcpy_h2d")
        _chk(self.L.crc_plain_expand(self.c, self.p(d_compact), values.size, self.p(d_plain), self.stream), "crc_plain_expand")
        self.sync()             # cp is the source of an asynchronous copy until here
        return values.size

    def decode(self, plain):
        return self.L.crc_decode(self.c, _pu(np.ascontiguousarray(plain)))

    def keygen(self, seed):
        sk = np.zeros((self.k, self.n), dtype=np.uint64); pk = np.zeros((2, self.k, self.n), dtype=np.uint64)
        _chk(self.L.crc_keygen(self.c, seed, _pu(sk), _pu(pk)), "crc_keygen"); return sk, pk

    def gen_evk(self, seed, sk, dbc=16):
        evk = np.zeros(self.L.crc_evk_words(self.c, dbc), dtype=np.uint64)
        _chk(self.L.crc_gen_evk(self.c, seed, _pu(sk), dbc, _pu(evk)), "crc_gen_evk"); return evk

    def encrypt(self, pk, plains, seed):
        plains = np.ascontiguousarray(plains); lead = plains.shape[:-1]
        cnt = int(np.prod(lead)) if lead else 1
        ct = np.zeros((cnt, 2, self.k, self.n), dtype=np.uint64)
        _chk(self.L.crc_encrypt(self.c, _pu(pk), _pu(plains.reshape(cnt, self.n)), cnt, seed, _pu(ct)), "crc_encrypt")
        return ct.reshape(lead + (2, self.k, self.n))

    # key-based client side (ChaCha20 under a 256-bit key; `key` = 32 bytes, random_key() draws it from the OS)
    @staticmethod
    def _key(key):
        key = bytes(key)
        assert len(key) == 32
        return (ctypes.c_uint8 * 32).from_buffer_copy(key)

    def random_key(self):
        buf = (ctypes.c_uint8 * 32)()
        _chk(self.L.crc_random_key(buf), "crc_random_key"); return bytes(buf)

    def keygen_key(self, key):
        sk = np.zeros((self.k, self.n), dtype=np.uint64); pk = np.zeros((2, self.k, self.n), dtype=np.uint64)
        _chk(self.L.crc_keygen_key(self.c, self._key(key), _pu(sk), _pu(pk)), "crc_keygen_key"); return sk, pk

    def gen_evk_key(self, key, sk, dbc=16):
        evk = np.zeros(self.L.crc_evk_words(self.c, dbc), dtype=np.uint64)
        _chk(self.L.crc_gen_evk_key(self.c, self._key(key), _pu(sk), dbc, _pu(evk)), "crc_gen_evk_key"); return evk

    def encrypt_key(self, pk, plains, key, stream_base=0):
        plains = np.ascontiguousarray(plains); lead = plains.shape[:-1]
        cnt = int(np.prod(lead)) if lead else 1
        ct = np.zeros((cnt, 2, self.k, self.n), dtype=np.uint64)
        _chk(self.L.crc_encrypt_key(self.c, _pu(pk), _pu(plains.reshape(cnt, self.n)), cnt, self._key(key), stream_base, _pu(ct)), "crc_encrypt_key")
        return ct.reshape(lead + (2, self.k, self.n))

    def decrypt(self, sk, cts, size=2):
        cts = np.ascontiguousarray(cts); lead = cts.shape[:-3]
        cnt = int(np.prod(lead)) if lead else 1
        out = np.zeros((cnt, self.n), dtype=np.uint64)
        _chk(self.L.crc_decrypt(self.c, _pu(sk), _pu(cts), cnt, size, _pu(out)), "crc_decrypt")
        return out.reshape(lead + (self.n,))

    def noise_budget(self, sk, ct):
        ct = np.ascontiguousarray(ct)
        return _chk(self.L.crc_noise_budget(self.c, _pu(sk), _pu(ct), ct.shape[-3]), "crc_noise_budget")

    # ---- device ops on DBuf / raw pointers (p(x) accepts DBuf, int, or torch tensor with data_ptr)
    @staticmethod
    def p(x):
        if x is None:
            return None
        if isinstance(x, DBuf):
            return x.ptr
        if hasattr(x, "data_ptr"):
            return x.data_ptr()
        return int(x)

    def plain_to_ntt(self, d_plain, count, d_out):
        _chk(self.L.crc_plain_to_ntt(self.c, self.p(d_plain), count, self.p(d_out), self.stream), "crc_plain_to_ntt")

    def plain_to_delta(self, d_plain, count, form, d_out):
        _chk(self.L.crc_plain_to_delta(self.c, self.p(d_plain), count, form, self.p(d_out), self.stream), "crc_plain_to_delta")

    def ntt_fwd(self, d_ct, count, size=2):
        _chk(self.L.crc_ntt_fwd(self.c, self.p(d_ct), count, size, self.stream), "crc_ntt_fwd")

    def ntt_inv(self, d_ct, count, size=2):
        _chk(self.L.crc_ntt_inv(self.c, self.p(d_ct), count, size, self.stream), "crc_ntt_inv")

    def add(self, d_acc, d_b, count, size=2):
        _chk(self.L.crc_add(self.c, self.p(d_acc), self.p(d_b), count, size, self.stream), "crc_add")

    def add_plain(self, d_ct, d_delta, count, group, sign=1):
        _chk(self.L.crc_add_plain(self.c, self.p(d_ct), self.p(d_delta), count, group, sign, self.stream), "crc_add_plain")

    def multiply_plain_ntt(self, d_ct, d_w, count, group, size=2):
        _chk(self.L.crc_multiply_plain_ntt(self.c, self.p(d_ct), self.p(d_w), count, group, size, self.stream), "crc_multiply_plain_ntt")

    def multiply_plain(self, d_ct, d_w, count, group):
        _chk(self.L.crc_multiply_plain(self.c, self.p(d_ct), self.p(d_w), count, group, self.stream), "crc_multiply_plain")

    def conv2d_work_bytes(self, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form):
        return self.L.crc_conv2d_work_bytes(self.c, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form)

    def conv2d(self, d_x, d_w, d_bias, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form, out_form, d_y, d_work, w_form=NTT):
        if w_form == NTT and in_form != NTTP and out_form != NTTP:
            _chk(self.L.crc_conv2d(self.c, self.p(d_x), self.p(d_w), self.p(d_bias), B, zd, xd, yd, xs, ys, xf, yf, nf, in_form, out_form,
                                   self.p(d_y), self.p(d_work), self.stream), "crc_conv2d")
        else:
            _chk(self.L.crc_conv2d_forms(self.c, self.p(d_x), self.p(d_w), w_form, self.p(d_bias), B, zd, xd, yd, xs, ys, xf, yf, nf, in_form, out_form,
                                         self.p(d_y), self.p(d_work), self.stream), "crc_conv2d_forms")

    # ---- limb form (CRC_NTTL): conv / dense on the matrix cores
    def limb_supported(self, zd, xf=1, yf=1):
        return bool(self.L.crc_limb_supported(self.c, zd, xf, yf))

    def limb_tensor_bytes(self, B, zd, xd=1, yd=1):
        return self.L.crc_limb_tensor_bytes(self.c, B, zd, xd, yd)

    def limb_weights_bytes(self, nf, zd, xf=1, yf=1):
        return self.L.crc_limb_weights_bytes(self.c, nf, zd, xf, yf)

    def limb_pack_weights(self, d_w_ntt, nf, zd, xf, yf, d_wl):
        _chk(self.L.crc_limb_pack_weights(self.c, self.p(d_w_ntt), nf, zd, xf, yf, self.p(d_wl), self.stream), "crc_limb_pack_weights")

    def plan_mac(self, zd, xd, yd, xs, ys, xf, yf, nf, B, matrix_cores=True):
        """the weight form (= kernel) of a conv / dense layer launched on B images (B = 0: shape only): crc_plan_mac, the policy shared with the C++ host classes"""
        wf = CI(0)
        _chk(self.L.crc_plan_mac(self.c, zd, xd, yd, xs, ys, xf, yf, nf, int(B or 0), 1 if matrix_cores else 0, ctypes.byref(wf)), "crc_plan_mac")
        return wf.value

    def plan_fold_pool(self, zd, xd, yd, xs, ys, xf, yf, nf, pxs, pys, pxf, pyf):
        f = CI(0)
        _chk(self.L.crc_plan_fold_pool(self.c, zd, xd, yd, xs, ys, xf, yf, nf, pxs, pys, pxf, pyf, ctypes.byref(f)), "crc_plan_fold_pool")
        return bool(f.value)

    def limb_pack_weights_tile(self, d_w_tile, nf, f0, ft, zd, xf, yf, d_wl):
        _chk(self.L.crc_limb_pack_weights_tile(self.c, self.p(d_w_tile), nf, f0, ft, zd, xf, yf, self.p(d_wl), self.stream), "crc_limb_pack_weights_tile")

    def limb_pack_tensor(self, d_x, in_form, B, zd, xd, yd, d_xl):
        _chk(self.L.crc_limb_pack_tensor(self.c, self.p(d_x), in_form, B, zd, xd, yd, self.p(d_xl), self.stream), "crc_limb_pack_tensor")

    def limb_pack_tensor_at(self, d_x, in_form, B, zd, xd, yd, d_xl, Btot, b0):
        _chk(self.L.crc_limb_pack_tensor_at(self.c, self.p(d_x), in_form, B, zd, xd, yd, self.p(d_xl), Btot, b0, self.stream), "crc_limb_pack_tensor_at")

    # ---- one-channel convolutions on the matrix cores (weight form CRC_NTTL1)
    def limb_conv1_supported(self, zd, xd, yd, xs, ys, xf, yf, nf):
        return bool(self.L.crc_limb_conv1_supported(self.c, zd, xd, yd, xs, ys, xf, yf, nf))

    def limb_conv1_weights_bytes(self):
        return self.L.crc_limb_conv1_weights_bytes(self.c)

    def limb_conv1_pack_weights(self, d_w_ntt, nf, xf, yf, d_wl):
        _chk(self.L.crc_limb_conv1_pack_weights(self.c, self.p(d_w_ntt), nf, xf, yf, self.p(d_wl), self.stream), "crc_limb_conv1_pack_weights")

    def conv2d_forms_work_bytes(self, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form, w_form, out_form):
        return self.L.crc_conv2d_forms_work_bytes(self.c, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form, w_form, out_form)

    def pack28(self, d_rows, rows, unpack=False):
        _chk(self.L.crc_pack28(self.c, self.p(d_rows), rows, 1 if unpack else 0, self.stream), "crc_pack28")

    def conv2d_fold_pool(self, d_w, d_bias_ntt, d_div_ntt, nf, zd, xf, yf, cxs, cys, pxf, pyf, d_w_out, d_bias_out):
        _chk(self.L.crc_conv2d_fold_pool(self.c, self.p(d_w), self.p(d_bias_ntt), self.p(d_div_ntt), nf, zd, xf, yf, cxs, cys, pxf, pyf,
                                         self.p(d_w_out), self.p(d_bias_out), self.stream), "crc_conv2d_fold_pool")

    def dense_work_bytes(self, B, in_dim, out_dim, in_form):
        return self.L.crc_dense_work_bytes(self.c, B, in_dim, out_dim, in_form)

    def dense(self, d_x, d_w, d_bias, B, in_dim, out_dim, in_form, out_form, d_y, d_work, w_form=NTT):
        if w_form == NTT and in_form != NTTP and out_form != NTTP:
            _chk(self.L.crc_dense(self.c, self.p(d_x), self.p(d_w), self.p(d_bias), B, in_dim, out_dim, in_form, out_form, self.p(d_y), self.p(d_work),
                                  self.stream), "crc_dense")
        else:
            _chk(self.L.crc_dense_forms(self.c, self.p(d_x), self.p(d_w), w_form, self.p(d_bias), B, in_dim, out_dim, in_form, out_form, self.p(d_y),
                                        self.p(d_work), self.stream), "crc_dense_forms")

    def pool(self, d_x, B, zd, xd, yd, xs, ys, xf, yf, d_div, form, d_y):
        _chk(self.L.crc_pool(self.c, self.p(d_x), B, zd, xd, yd, xs, ys, xf, yf, self.p(d_div), form, self.p(d_y), self.stream), "crc_pool")

    def batchnorm(self, d_x, B, zd, xd, yd, d_mean, d_invstd, form):
        _chk(self.L.crc_batchnorm(self.c, self.p(d_x), B, zd, xd, yd, self.p(d_mean), self.p(d_invstd), form, self.stream), "crc_batchnorm")

    def square_relin_work_bytes(self, count, dbc=16):
        return self.L.crc_square_relin_work_bytes(self.c, count, dbc)

    def square_relin(self, d_x, count, d_evk, d_y, d_work, dbc=16, in_form=COEFF, out_form=COEFF):
        _chk(self.L.crc_square_relin_forms(self.c, self.p(d_x), in_form, count, self.p(d_evk), dbc, self.p(d_y), out_form, self.p(d_work), self.stream), "crc_square_relin_forms")

    def square_pool_relin_supported(self, xf, yf, dbc=16):
        return bool(self.L.crc_square_pool_relin_supported(self.c, dbc, xf, yf))

    def square_pool_relin_work_bytes(self, B, zd, xd, yd, xs, ys, xf, yf, dbc=16):
        return self.L.crc_square_pool_relin_work_bytes(self.c, B, zd, xd, yd, xs, ys, xf, yf, dbc)

    def square_pool_relin(self, d_x, B, zd, xd, yd, xs, ys, xf, yf, d_evk, d_y, d_work, dbc=16, in_form=COEFF, out_form=COEFF, d_div=None):
        """Square + relinearise + sum / average pooling with one key switch per pooled ciphertext (crc_square_pool_relin_forms)"""
        _chk(self.L.crc_square_pool_relin_forms(self.c, self.p(d_x), in_form, B, zd, xd, yd, xs, ys, xf, yf, self.p(d_evk), dbc, self.p(d_div), self.p(d_y), out_form, self.p(d_work),
                                                self.stream),
             "crc_square_pool_relin_forms")

    def encrypt_dev_work_bytes(self, count):
        return self.L.crc_encrypt_dev_work_bytes(self.c, count)

    def encrypt_dev(self, d_pk, d_plain, count, seed, d_ct, d_work):
        _chk(self.L.crc_encrypt_dev(self.c, self.p(d_pk), self.p(d_plain), count, seed, self.p(d_ct), self.p(d_work), self.stream), "crc_encrypt_dev")

    def encrypt_dev_forms(self, d_pk, d_plain, count, seed, out_form, d_ct, d_work):
        _chk(self.L.crc_encrypt_dev_forms(self.c, self.p(d_pk), self.p(d_plain), count, seed, out_form, self.p(d_ct), self.p(d_work), self.stream),
             "crc_encrypt_dev_forms")

    def encrypt_dev_key_forms(self, d_pk, d_plain, count, key, stream_base, out_form, d_ct, d_work):
        _chk(self.L.crc_encrypt_dev_key_forms(self.c, self.p(d_pk), self.p(d_plain), count, self._key(key), stream_base, out_form, self.p(d_ct),
                                              self.p(d_work), self.stream), "crc_encrypt_dev_key_forms")

    # ---- Decryptor::decrypt / FractionalEncoder / the refresh of Network::forward on the device (kernels_decrypt.hip)
    def decrypt_dev_work_bytes(self, count, size=2, in_form=COEFF):
        return self.L.crc_decrypt_dev_work_bytes(self.c, count, size, in_form)

    def decrypt_dev(self, d_sk, d_ct, count, d_plain, d_work, size=2, in_form=COEFF):
        _chk(self.L.crc_decrypt_dev(self.c, self.p(d_sk), self.p(d_ct), count, size, in_form, self.p(d_plain), self.p(d_work), self.stream), "crc_decrypt_dev")

    def decode_dev(self, d_plain, count, d_out):
        _chk(self.L.crc_decode_dev(self.c, self.p(d_plain), count, self.p(d_out), self.stream), "crc_decode_dev")

    def encode_dev(self, d_values, count, d_plain, f64=False):
        f = self.L.crc_encode_dev_f64 if f64 else self.L.crc_encode_dev_f32
        _chk(f(self.c, self.p(d_values), count, self.p(d_plain), self.stream), "crc_encode_dev")

    def refresh_dev_work_bytes(self, count, in_form=COEFF):
        return self.L.crc_refresh_dev_work_bytes(self.c, count, in_form)

    def refresh_dev(self, d_sk, d_pk, d_ct_in, count, seed, d_ct_out, d_work, in_form=COEFF, out_form=COEFF, d_values=None, key=None, stream_base=0):
        if key is None:
            _chk(self.L.crc_refresh_dev(self.c, self.p(d_sk), self.p(d_pk), self.p(d_ct_in), count, in_form, seed, out_form, self.p(d_ct_out), self.p(d_values),
                                        self.p(d_work), self.stream), "crc_refresh_dev")
        else:
            _chk(self.L.crc_refresh_dev_key(self.c, self.p(d_sk), self.p(d_pk), self.p(d_ct_in), count, in_form, self._key(key), stream_base, out_form,
                                            self.p(d_ct_out), self.p(d_values), self.p(d_work), self.stream), "crc_refresh_dev_key")

    def encrypt_dev_noise_thresholds(self):
        out = (ctypes.c_uint64 * 19)()
        self.L.crc_encrypt_dev_noise_thresholds(out)
        return [int(v) for v in out]

    def encrypt_dev_key(self, d_pk, d_plain, count, key, stream_base, d_ct, d_work):
        _chk(self.L.crc_encrypt_dev_key(self.c, self.p(d_pk), self.p(d_plain), count, self._key(key), stream_base, self.p(d_ct), self.p(d_work), self.stream), "crc_encrypt_dev_key")

    # ---- multi-GPU: RCCL communicator bound to this context's device
    def comm_unique_id(self):
        buf = (ctypes.c_uint8 * 128)()
        _chk(self.L.crc_comm_unique_id(buf), "crc_comm_unique_id"); return bytes(buf)

    def comm_create(self, world, rank, uid):
        assert len(uid) == 128
        out = VP()
        _chk(self.L.crc_comm_create(self.c, world, rank, (ctypes.c_uint8 * 128).from_buffer_copy(bytes(uid)), ctypes.byref(out)), "crc_comm_create")
        return out

    def comm_destroy(self, comm):
        self.L.crc_comm_destroy(comm)

    def broadcast_weights(self, comm, d_w, nbytes, root=0):
        assert nbytes % 8 == 0
        _chk(self.L.crc_broadcast_weights(comm, self.p(d_w), nbytes // 8, root, self.stream), "crc_broadcast_weights")

    def allgather_u64(self, comm, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros((self.L.crc_comm_world(comm), v.size), dtype=np.uint64)
        _chk(self.L.crc_comm_allgather_u64(comm, _pu(v), v.size, _pu(out), self.stream), "crc_comm_allgather_u64")
        return out

    def checksum64(self, d_words, nbytes):
        out = np.zeros(2, dtype=np.uint64)
        _chk(self.L.crc_checksum64(self.c, self.p(d_words), nbytes // 8, _pu(out), self.stream), "crc_checksum64")
        return int(out[0]), int(out[1])

    def square(self, d_x, count, d_y3, d_work):
        _chk(self.L.crc_square(self.c, self.p(d_x), count, self.p(d_y3), self.p(d_work), self.stream), "crc_square")

    def relinearize(self, d_x3, count, d_evk, d_y, d_work, dbc=16):
        _chk(self.L.crc_relinearize(self.c, self.p(d_x3), count, self.p(d_evk), dbc, self.p(d_y), self.p(d_work), self.stream), "crc_relinearize")
