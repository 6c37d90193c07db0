"""Client-side randomness of the product (keys and encryption noise must come from a CSPRNG keyed with OS entropy).
CPU only: the ChaCha20 block function against RFC 8439's known answer and an independent pure-Python restatement; the sampling
laws (uniform ternary, clipped normal sigma 3.19 cut at 6 sigma -- SEAL util/globals.cpp:13-15) observed through public
structure; key-based entry points are key- and stream-separated."""
import ctypes
import os
import struct

import numpy as np

import crcnn_amd as ca
from crcnn_amd import binding

Q = [0x7fffffff380001, 0x3fffffff000001]


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & 0xffffffff


def _chacha_block_py(key, counter, nonce):
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(struct.unpack("<8I", key)) + [counter] + list(struct.unpack("<3I", nonce))
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = _rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(x[i] + st[i]) & 0xffffffff for i in range(16)])


def _block(key, counter, nonce):
    L = binding.load()
    out = (ctypes.c_uint8 * 64)()
    assert L.crc_chacha20_block((ctypes.c_uint8 * 32).from_buffer_copy(key), counter, (ctypes.c_uint8 * 12).from_buffer_copy(nonce), out) == 0
    return bytes(out)


def test_chacha20_block_known_answer():
    key = bytes(range(32)); nonce = bytes.fromhex("000000090000004a00000000")
    want = bytes.fromhex("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")       # RFC 8439, 2.3.2
    assert _chacha_block_py(key, 1, nonce) == want
    assert _block(key, 1, nonce) == want
    rng = np.random.default_rng(1)
    for _ in range(20):
        k = rng.bytes(32); n = rng.bytes(12); c = int(rng.integers(0, 1 << 32))
        assert _block(k, c, n) == _chacha_block_py(k, c, n)


def test_random_key_comes_from_the_os_and_separates_everything():
    E = ca.Engine(1024, Q, 1 << 20, device=-1)
    k1, k2 = E.random_key(), E.random_key()
    assert k1 != k2 and len(k1) == 32 and len(set(k1)) > 8
    sk1, pk1 = E.keygen_key(k1); sk1b, pk1b = E.keygen_key(k1); sk2, pk2 = E.keygen_key(k2)
    assert np.array_equal(sk1, sk1b) and np.array_equal(pk1, pk1b)
    assert not np.array_equal(sk1, sk2) and not np.array_equal(pk1[1], pk2[1])
    pl, _ = E.encode(np.array([0.75, -2.5, 0.0], dtype=np.float32))
    a = E.encrypt_key(pk1, pl, k1, 0); b = E.encrypt_key(pk1, pl, k1, 3); c = E.encrypt_key(pk1, pl, k2, 0)
    assert np.array_equal(E.decrypt(sk1, a), pl) and np.array_equal(E.decrypt(sk1, b), pl)
    assert not np.array_equal(a[:, 1], b[:, 1]) and not np.array_equal(a[:, 1], c[:, 1])
    assert np.array_equal(b[0, 1], E.encrypt_key(pk1, pl, k1, 2)[1, 1])          # stream id = stream_base + ciphertext index
    # the seeded entry points are deterministic and unrelated to each other across seeds
    s1 = E.keygen(5)[0]; assert np.array_equal(s1, E.keygen(5)[0]) and not np.array_equal(s1, E.keygen(6)[0])


def test_sampling_laws():
    """secret key: uniform over {-1, 0, 1} (read back by decrypting the noiseless "ciphertext" (0, Delta): c0 + c1 s = Delta s); fresh
    ciphertexts: the noise budget SEAL measures for the same parameters (78 bits at n=4096, k=2, t=2^20: SURVEY 8c), concentrated"""
    n, t = 4096, 1 << 20
    E = ca.Engine(n, Q, t, device=-1)
    key = E.random_key()
    sk_ntt, pk = E.keygen_key(key)
    ct = np.zeros((1, 2, len(Q), n), dtype=np.uint64)
    ct[0, 1, :, 0] = E.table("delta")
    s = E.decrypt(sk_ntt, ct)[0]
    counts = [int(np.sum(s == v)) for v in (0, 1, t - 1)]
    assert sum(counts) == n, "secret key is not ternary"
    assert all(abs(c - n / 3) < 5 * np.sqrt(n * 2 / 9) for c in counts), counts
    pl = np.zeros((16, n), dtype=np.uint64)
    cts = E.encrypt_key(pk, pl, key, 0)
    b = [E.noise_budget(sk_ntt, cts[i]) for i in range(16)]
    assert max(b) - min(b) <= 2 and 76 <= min(b) <= 80, b
    assert not np.any(E.decrypt(sk_ntt, cts))


def test_host_threads_do_not_change_the_bits():
    """csrc/host_parallel.h: encoding and encryption run item ranges on CRC_HOST_THREADS threads (one keystream per ciphertext, one weight per plaintext):
    1 thread and 7 threads give the same plaintexts and ciphertexts"""
    import hashlib
    import subprocess
    import sys
    prog = ("import numpy as np, hashlib, crcnn_amd as ca\n"
            "n = 1024; q = ca.default_coeff_modulus_128(n)[:1]\n"
            "E = ca.Engine(n, q, 1 << 20, device=-1)\n"
            "sk, pk = E.keygen(11)\n"
            "vals = np.random.default_rng(5).standard_normal(300).astype(np.float32)\n"
            "pl, cc = E.encode(vals)\n"
            "ct = E.encrypt(pk, pl, 99)\n"
            "print(hashlib.sha256(pl).hexdigest(), hashlib.sha256(cc).hexdigest(), hashlib.sha256(ct).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [subprocess.check_output([sys.executable, "-c", prog], cwd=root, env=dict(os.environ, CRC_HOST_THREADS=str(t), PYTHONPATH=root), text=True).strip() for t in (1, 7)]
    assert outs[0] == outs[1] and len(outs[0].split()) == 3
