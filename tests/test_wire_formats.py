"""SEAL 2.3.1 wire formats (crcnn_amd/csrc/wire.cpp) byte-for-byte against the bytes SEAL itself wrote for the same
objects (tests/golden/ops_n256_*.npz: `Ciphertext::save`, `EvaluationKeys::save`, `PublicKey::save`, `SecretKey::save`
captured by oracle/ref_harness.cpp), plus the SHA3-256 parameter hash (encryptionparams.cpp:69-100).  CPU only."""
import ctypes
import glob
import os

import numpy as np
import pytest

import crcnn_amd as ca

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SETS = sorted(glob.glob(os.path.join(GOLD, "ops_n256_*.npz")))
PU = ctypes.POINTER(ctypes.c_uint64)
SZ = ctypes.c_size_t


def P(a):
    return a.ctypes.data_as(PU)


@pytest.mark.parametrize("path", SETS + [os.path.join(GOLD, "ops_n2048_k1_t18.npz")], ids=lambda p: os.path.basename(p)[:-4])
def test_parameter_hash(path):
    g = dict(np.load(path))
    E = ca.Engine(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]), device=-1)
    out = np.zeros(4, dtype=np.uint64)
    assert E.L.crc_params_hash(E.c, P(out)) == 0
    assert np.array_equal(out, g["ref_params_hash"])


def test_known_hash_anchor():
    # (n=4096, coeff_modulus_128(4096), t=2^20)
    E = ca.Engine(4096, ca.default_coeff_modulus_128(4096), 1 << 20, device=-1)
    out = np.zeros(4, dtype=np.uint64)
    E.L.crc_params_hash(E.c, P(out))
    assert [hex(int(v)) for v in out] == ["0x5fd9fc79662ced93", "0xb8d15e8734b67360", "0xf363748f2b9c827e", "0x434c52819d637797"]


@pytest.mark.parametrize("path", SETS, ids=lambda p: os.path.basename(p)[:-4])
def test_save_equals_seal_bytes_and_load_roundtrips(path):
    g = dict(np.load(path))
    n, k = int(g["n"]), len(g["q"])
    E = ca.Engine(n, [int(x) for x in g["q"]], int(g["t"]), device=-1)
    L = E.L
    for f in ("crc_seal_ct_bytes", "crc_seal_evk_bytes", "crc_seal_pk_bytes", "crc_seal_sk_bytes"):
        getattr(L, f).restype = SZ
    w = SZ(0)

    def save(fn, arr, *extra):
        nbytes = {"ct": L.crc_seal_ct_bytes(E.c, 2), "evk": L.crc_seal_evk_bytes(E.c, 16), "pk": L.crc_seal_pk_bytes(E.c), "sk": L.crc_seal_sk_bytes(E.c)}[fn]
        buf = np.zeros(nbytes, dtype=np.uint8)
        rc = getattr(L, f"crc_seal_{fn}_save")(E.c, P(np.ascontiguousarray(arr)), *extra, buf.ctypes.data_as(ctypes.c_void_p), SZ(nbytes), ctypes.byref(w))
        assert rc == 0 and w.value == nbytes
        return buf

    ct = np.ascontiguousarray(g["ct_in"][0])
    b = save("ct", ct, 2); assert np.array_equal(b, g["ref_wire_ct"])
    back = np.zeros_like(ct); size = ctypes.c_int(0)
    assert L.crc_seal_ct_load(E.c, b.ctypes.data_as(ctypes.c_void_p), SZ(b.size), P(back), 2, ctypes.byref(size), None) == 0
    assert size.value == 2 and np.array_equal(back, ct)
    b = save("evk", g["evk"], 16); assert np.array_equal(b, g["ref_wire_evk"])
    back = np.zeros_like(g["evk"]); dbc = ctypes.c_int(0)
    assert L.crc_seal_evk_load(E.c, b.ctypes.data_as(ctypes.c_void_p), SZ(b.size), P(back), ctypes.byref(dbc)) == 0
    assert dbc.value == 16 and np.array_equal(back, g["evk"])
    b = save("pk", g["pk"]); assert np.array_equal(b, g["ref_wire_pk"])
    back = np.zeros_like(g["pk"]); assert L.crc_seal_pk_load(E.c, b.ctypes.data_as(ctypes.c_void_p), SZ(b.size), P(back)) == 0 and np.array_equal(back, g["pk"])
    b = save("sk", g["sk"]); assert np.array_equal(b, g["ref_wire_sk"])
    back = np.zeros_like(g["sk"]); assert L.crc_seal_sk_load(E.c, b.ctypes.data_as(ctypes.c_void_p), SZ(b.size), P(back)) == 0 and np.array_equal(back, g["sk"])
    # an object made for other parameters is rejected, as SEAL does ("encrypted is not valid for encryption parameters")
    E2 = ca.Engine(n, [int(x) for x in g["q"]], int(g["t"]) * 2, device=-1)
    bad = g["ref_wire_ct"]; tmp = np.zeros_like(ct)
    assert E2.L.crc_seal_ct_load(E2.c, bad.ctypes.data_as(ctypes.c_void_p), SZ(bad.size), P(tmp), 2, ctypes.byref(size), None) < 0
    assert L.crc_seal_ct_load(E.c, bad.ctypes.data_as(ctypes.c_void_p), SZ(bad.size - 8), P(tmp), 2, ctypes.byref(size), None) < 0      # truncated
