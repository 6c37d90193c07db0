"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(crcnn_amd) never does.  Arrays are numpy uint64, C-contiguous, layout [..][k][n] (see crc_oracle.h).
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
u64 = ctypes.c_uint64
PU = ctypes.POINTER(u64)
VP = ctypes.c_void_p
CI = ctypes.c_int

# SEAL/util/globals.cpp:25-90 (coeff_modulus_128 tables)
COEFF_MODULUS_128 = {
    1024: [0x7e00001],
    2048: [0x3fffffff000001],
    4096: [0x7fffffff380001, 0x3fffffff000001],
    8192: [0x7fffffff380001, 0x7ffffffef00001, 0x3fffffff000001, 0x3ffffffef40001],
    16384: [0x7fffffff380001, 0x7ffffffef00001, 0x7ffffffeac0001, 0x7ffffffe700001,
            0x7ffffffe600001, 0x7ffffffe4c0001, 0x3fffffff000001, 0x3ffffffef40001],
}


def build():
    """(re)build liboracle.so with gcc if missing or stale."""
    src = [os.path.join(_HERE, f) for f in ("crc_oracle.c", "crc_oracle.h")]
    if not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(PU)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_ctx_create.restype = VP
        L.orc_ctx_create.argtypes = [CI, PU, CI, u64]
        L.orc_ctx_destroy.argtypes = [VP]
        L.orc_ctx_table.argtypes = [VP, ctypes.c_char_p, PU, CI]
        L.orc_ctx_kbsk.argtypes = [VP]
        L.orc_ctx_evk_words.argtypes = [VP, CI]
        L.orc_const_ratio.argtypes = [u64, PU]
        L.orc_barrett_reduce_128.restype = u64
        L.orc_barrett_reduce_128.argtypes = [u64, u64, u64]
        L.orc_mulmod.restype = u64
        L.orc_mulmod.argtypes = [u64, u64, u64]
        L.orc_min_primitive_root.restype = u64
        L.orc_min_primitive_root.argtypes = [u64, u64]
        L.orc_ntt_fwd.argtypes = [VP, CI, PU]
        L.orc_ntt_inv.argtypes = [VP, CI, PU]
        L.orc_dyadic.argtypes = [VP, CI, PU, PU, PU]
        L.orc_encode.argtypes = [VP, ctypes.c_double, PU]
        L.orc_decode.restype = ctypes.c_double
        L.orc_decode.argtypes = [VP, PU]
        L.orc_plain_to_ntt.argtypes = [VP, PU, PU]
        L.orc_ct_to_ntt.argtypes = [VP, PU, CI]
        L.orc_ct_from_ntt.argtypes = [VP, PU, CI]
        L.orc_multiply_plain_ntt.argtypes = [VP, PU, CI, PU]
        L.orc_add.argtypes = [VP, PU, PU, CI]
        L.orc_add_plain.argtypes = [VP, PU, PU]
        L.orc_sub_plain.argtypes = [VP, PU, PU]
        L.orc_multiply_plain.argtypes = [VP, PU, CI, PU]
        L.orc_square.argtypes = [VP, PU, PU]
        L.orc_relinearize.argtypes = [VP, PU, PU, CI, PU]
        L.orc_keygen.argtypes = [VP, u64, PU, PU]
        L.orc_gen_evk.argtypes = [VP, u64, PU, CI, PU]
        L.orc_encrypt.argtypes = [VP, PU, PU, u64, PU]
        L.orc_decrypt.argtypes = [VP, PU, PU, CI, PU]
        L.orc_noise_budget.argtypes = [VP, PU, PU, CI]
        L.orc_conv_forward.argtypes = [VP, PU] + [CI] * 8 + [PU, PU, PU, CI, CI, CI]
        L.orc_conv_forward_fast.argtypes = [VP, PU] + [CI] * 8 + [PU, PU, PU, CI]
        L.orc_fc_forward.argtypes = [VP, PU, CI, CI, PU, PU, PU, CI, CI, CI]
        L.orc_pool_forward.argtypes = [VP, PU] + [CI] * 7 + [PU, PU, CI]
        L.orc_bn_forward.argtypes = [VP, PU, CI, CI, CI, PU, PU, CI]
        L.orc_square_forward.argtypes = [VP, PU, ctypes.c_size_t, PU, CI, PU, CI]
        _lib = L
    return _lib


class Oracle:
    """One (n, q[], t) parameter set."""

    def __init__(self, n, q, t):
        self.L = lib()
        self.n, self.k, self.t = int(n), len(q), int(t)
        self.q = np.array(q, dtype=np.uint64)
        self.c = self.L.orc_ctx_create(self.n, _p(self.q), self.k, self.t)
        if not self.c:
            raise ValueError("invalid parameters")
        self.kbsk = self.L.orc_ctx_kbsk(self.c)

    def __del__(self):
        try:
            self.L.orc_ctx_destroy(self.c)
        except Exception:
            pass

    # ---- tables
    def table(self, name, cap=1 << 16):
        out = np.zeros(cap, dtype=np.uint64)
        cnt = self.L.orc_ctx_table(self.c, name.encode(), _p(out), cap)
        if cnt < 0:
            raise KeyError(name)
        return out[:cnt].copy()

    # ---- shapes
    def ct(self, *lead, size=2):
        return np.zeros(tuple(lead) + (size, self.k, self.n), dtype=np.uint64)

    # ---- encoder
    def encode(self, v):
        out = np.zeros(self.n, dtype=np.uint64)
        cc = self.L.orc_encode(self.c, float(v), _p(out))
        return out, cc

    def encode_many(self, vals):
        vals = np.asarray(vals).reshape(-1)
        out = np.zeros((len(vals), self.n), dtype=np.uint64)
        for i, v in enumerate(vals):
            self.L.orc_encode(self.c, float(v), _p(out[i]))
        return out

    def decode(self, coeffs):
        return self.L.orc_decode(self.c, _p(np.ascontiguousarray(coeffs)))

    # ---- ops (in place on copies)
    def ntt_fwd(self, mi, poly):
        a = np.ascontiguousarray(poly).copy(); self.L.orc_ntt_fwd(self.c, mi, _p(a)); return a

    def ntt_inv(self, mi, poly):
        a = np.ascontiguousarray(poly).copy(); self.L.orc_ntt_inv(self.c, mi, _p(a)); return a

    def plain_to_ntt(self, plain):
        out = np.zeros((self.k, self.n), dtype=np.uint64)
        self.L.orc_plain_to_ntt(self.c, _p(np.ascontiguousarray(plain)), _p(out)); return out

    def plains_to_ntt(self, plains):
        plains = np.ascontiguousarray(plains)
        lead = plains.shape[:-1]
        flat = plains.reshape(-1, self.n)
        out = np.zeros((flat.shape[0], self.k, self.n), dtype=np.uint64)
        for i in range(flat.shape[0]):
            self.L.orc_plain_to_ntt(self.c, _p(flat[i]), _p(out[i]))
        return out.reshape(lead + (self.k, self.n))

    def ct_to_ntt(self, ct):
        a = np.ascontiguousarray(ct).copy(); self.L.orc_ct_to_ntt(self.c, _p(a), a.shape[-3]); return a

    def ct_from_ntt(self, ct):
        a = np.ascontiguousarray(ct).copy(); self.L.orc_ct_from_ntt(self.c, _p(a), a.shape[-3]); return a

    def multiply_plain_ntt(self, ct, w):
        a = np.ascontiguousarray(ct).copy(); self.L.orc_multiply_plain_ntt(self.c, _p(a), a.shape[-3], _p(np.ascontiguousarray(w))); return a

    def add(self, a, b):
        r = np.ascontiguousarray(a).copy(); self.L.orc_add(self.c, _p(r), _p(np.ascontiguousarray(b)), r.shape[-3]); return r

    def add_plain(self, ct, plain):
        r = np.ascontiguousarray(ct).copy(); self.L.orc_add_plain(self.c, _p(r), _p(np.ascontiguousarray(plain))); return r

    def sub_plain(self, ct, plain):
        r = np.ascontiguousarray(ct).copy(); self.L.orc_sub_plain(self.c, _p(r), _p(np.ascontiguousarray(plain))); return r

    def multiply_plain(self, ct, plain):
        r = np.ascontiguousarray(ct).copy(); self.L.orc_multiply_plain(self.c, _p(r), r.shape[-3], _p(np.ascontiguousarray(plain))); return r

    def square(self, ct):
        out = self.ct(size=3); self.L.orc_square(self.c, _p(np.ascontiguousarray(ct)), _p(out)); return out

    def relinearize(self, ct3, evk, dbc=16):
        out = self.ct(); self.L.orc_relinearize(self.c, _p(np.ascontiguousarray(ct3)), _p(evk), dbc, _p(out)); return out

    # ---- client side
    def keygen(self, seed):
        sk = np.zeros((self.k, self.n), dtype=np.uint64); pk = self.ct()
        self.L.orc_keygen(self.c, seed, _p(sk), _p(pk)); return sk, pk

    def gen_evk(self, seed, sk, dbc=16):
        evk = np.zeros(self.L.orc_ctx_evk_words(self.c, dbc), dtype=np.uint64)
        self.L.orc_gen_evk(self.c, seed, _p(sk), dbc, _p(evk)); return evk

    def encrypt(self, pk, plain, seed):
        ct = self.ct(); self.L.orc_encrypt(self.c, _p(pk), _p(np.ascontiguousarray(plain)), seed, _p(ct)); return ct

    def encrypt_many(self, pk, plains, seed0):
        plains = np.ascontiguousarray(plains); lead = plains.shape[:-1]
        flat = plains.reshape(-1, self.n)
        out = self.ct(flat.shape[0])
        for i in range(flat.shape[0]):
            self.L.orc_encrypt(self.c, _p(pk), _p(flat[i]), seed0 + i, _p(out[i]))
        return out.reshape(lead + (2, self.k, self.n))

    def decrypt(self, sk, ct):
        ct = np.ascontiguousarray(ct); out = np.zeros(self.n, dtype=np.uint64)
        self.L.orc_decrypt(self.c, _p(sk), _p(ct), ct.shape[-3], _p(out)); return out

    def decrypt_value(self, sk, ct):
        return self.decode(self.decrypt(sk, ct))

    def noise_budget(self, sk, ct):
        ct = np.ascontiguousarray(ct)
        return self.L.orc_noise_budget(self.c, _p(sk), _p(ct), ct.shape[-3])

    # ---- layers (reference operation order)
    def conv(self, x, w_ntt, bias_plain, xs, ys, threads=1, fast=False, f_range=None):
        zd, xd, yd = x.shape[:3]; nf, _, xf, yf = w_ntt.shape[:4]
        xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
        y = self.ct(nf, xo, yo)
        x = np.ascontiguousarray(x); w_ntt = np.ascontiguousarray(w_ntt); bias_plain = np.ascontiguousarray(bias_plain)
        if fast:
            self.L.orc_conv_forward_fast(self.c, _p(x), zd, xd, yd, xs, ys, xf, yf, nf, _p(w_ntt), _p(bias_plain), _p(y), threads)
        else:
            fb, fe = f_range if f_range else (0, nf)
            self.L.orc_conv_forward(self.c, _p(x), zd, xd, yd, xs, ys, xf, yf, nf, _p(w_ntt), _p(bias_plain), _p(y), threads, fb, fe)
        return y

    def fc(self, x, w_ntt, bias_plain, threads=1, r_range=None):
        out_dim, in_dim = w_ntt.shape[:2]
        x = np.ascontiguousarray(x).reshape(in_dim, 2, self.k, self.n)
        y = self.ct(out_dim)
        rb, re = r_range if r_range else (0, out_dim)
        self.L.orc_fc_forward(self.c, _p(x), in_dim, out_dim, _p(np.ascontiguousarray(w_ntt)), _p(np.ascontiguousarray(bias_plain)), _p(y), threads, rb, re)
        return y.reshape(1, out_dim, 1, 2, self.k, self.n)

    def pool(self, x, xs, ys, xf, yf, div_plain=None, threads=1):
        zd, xd, yd = x.shape[:3]
        xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
        y = self.ct(zd, xo, yo)
        dp = _p(np.ascontiguousarray(div_plain)) if div_plain is not None else None
        self.L.orc_pool_forward(self.c, _p(np.ascontiguousarray(x)), zd, xd, yd, xs, ys, xf, yf, dp, _p(y), threads)
        return y

    def bn(self, x, mean_plain, invstd_plain, threads=1):
        r = np.ascontiguousarray(x).copy(); zd, xd, yd = r.shape[:3]
        self.L.orc_bn_forward(self.c, _p(r), zd, xd, yd, _p(np.ascontiguousarray(mean_plain)), _p(np.ascontiguousarray(invstd_plain)), threads)
        return r

    def square_layer(self, x, evk, dbc=16, threads=1):
        x = np.ascontiguousarray(x); y = np.zeros_like(x)
        cnt = int(np.prod(x.shape[:-3]))
        self.L.orc_square_forward(self.c, _p(x), cnt, _p(evk), dbc, _p(y), threads)
        return y


def synth_image(index, seed=0xC0FFEE):
    """Synthetic MNIST-like 28x28 uint8 image: 81% zeros, rest uniform 1..255, splitmix64 stream."""
    mask = (1 << 64) - 1
    s = (seed + index) & mask
    out = np.zeros(784, dtype=np.uint8)
    for i in range(784):
        s = (s + 0x9E3779B97F4A7C15) & mask
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        if (z & 0xFFFF) >= int(0.81 * 65536):
            out[i] = 1 + ((z >> 16) % 255)
    return out.reshape(28, 28)


def normalize(img_u8):
    """CrCNN/src/utils.cpp:9-18,27: float32 (p/255 - 0.1307)/0.3081."""
    p = img_u8.astype(np.float32)
    return ((p / np.float32(255)) - np.float32(0.1307)) / np.float32(0.3081)
