"""Known-answer vectors written down by the reference's own authors (SEAL 2.3.1's SEALTest project), replayed on the CPU oracle.

SURVEY section 4 lists them as the only vectors the reference *holds* for this path (everything else in tests/golden was produced by running
the compiled reference).  Sources, values copied as data:
  SEALTest/util/smallntt.cpp:52-102          root-power tables of q = 0xffffffffffc0001 at n = 2 and n = 4; NTT of [0,0], [1,0], [1,1] at n = 2
  SEALTest/util/smallntt.cpp:104-133         inverse(forward(x)) = x at n = 8
  SEALTest/util/uintarithsmallmod.cpp:143-213 barrett_reduce_128 and multiply_uint_uint_mod
The product's own modular arithmetic header replays the same arithmetic vectors in tests/cpp/modarith_check.cpp (test_modarith_cpu.py); its transforms
start at n = 64 and are pinned to this oracle (tests/test_abi_cpu.py, tests/test_gpu_ops.py).
SEALTest/baseconverter.cpp:46-548 is NOT usable: it takes its moduli from a `primes.h` that the reference tree does not contain (and constructs
BaseConverter with a signature 2.3.1 no longer has), so the primes behind its expected values are unknown; the base converter is pinned by the
compiled reference's square outputs instead (tests/golden/ops_*.npz ref_sq)."""
import ctypes

import numpy as np

from oracle import orc

Q60 = 0xffffffffffc0001


def test_smallntt_primitive_root_tables():
    """SmallNTTPrimitiveRootsTest, SEALTest/util/smallntt.cpp:52-72"""
    O = orc.Oracle(2, [Q60], 1 << 4)
    rp = [int(v) for v in O.table("root_powers:0")]
    assert rp == [1, 288794978602139552]
    O = orc.Oracle(4, [Q60], 1 << 4)
    rp = [int(v) for v in O.table("root_powers:0")]
    assert rp == [1, 288794978602139552, 178930308976060547, 748001537669050592]
    # get_from_inv_root_powers(1) == root_powers(1)^-1 (:61-63); the oracle keeps SEAL's inv_root_powers_div_two = that / 2
    inv = pow(rp[1], -1, Q60)
    half = pow(2, -1, Q60)
    assert int(O.table("inv_root_powers_div_two:0")[1]) == inv * half % Q60


def test_negacyclic_ntt_known_answers():
    """NegacyclicSmallNTTTest, SEALTest/util/smallntt.cpp:74-102"""
    O = orc.Oracle(2, [Q60], 1 << 4)
    assert [int(v) for v in O.ntt_fwd(0, np.array([0, 0], dtype=np.uint64))] == [0, 0]
    assert [int(v) for v in O.ntt_fwd(0, np.array([1, 0], dtype=np.uint64))] == [1, 1]
    assert [int(v) for v in O.ntt_fwd(0, np.array([1, 1], dtype=np.uint64))] == [288794978602139553, 864126526004445282]


def test_inverse_negacyclic_ntt_round_trip():
    """InverseNegacyclicSmallNTTTest, SEALTest/util/smallntt.cpp:104-133 (n = 8; the reference draws its input from random_device)"""
    O = orc.Oracle(8, [Q60], 1 << 4)
    z = np.zeros(8, dtype=np.uint64)
    assert not O.ntt_inv(0, z).any()
    rng = np.random.default_rng(5)
    for _ in range(100):
        x = rng.integers(0, Q60, size=8, dtype=np.uint64)
        assert np.array_equal(O.ntt_inv(0, O.ntt_fwd(0, x.copy())), x)


def _lib():
    L = orc.lib()
    L.orc_barrett_reduce_128.restype = ctypes.c_uint64
    L.orc_barrett_reduce_128.argtypes = [ctypes.c_uint64] * 3
    L.orc_mulmod.restype = ctypes.c_uint64
    L.orc_mulmod.argtypes = [ctypes.c_uint64] * 3
    return L


ALL = 0xFFFFFFFFFFFFFFFF
BARRETT = [  # (lo, hi, modulus, expected)  SEALTest/util/uintarithsmallmod.cpp:143-184
    (0, 0, 2, 0), (1, 0, 2, 1), (ALL, ALL, 2, 1),
    (0, 0, 3, 0), (1, 0, 3, 1), (123, 456, 3, 0), (ALL, ALL, 3, 0),
    (0, 0, 13131313131313, 0), (1, 0, 13131313131313, 1), (123, 456, 13131313131313, 8722750765283), (24242424242424, 79797979797979, 13131313131313, 1010101010101),
]
M62 = 4611686018427289601
MULMOD = [  # (a, b, modulus, expected)  SEALTest/util/uintarithsmallmod.cpp:186-213
    (0, 0, 2, 0), (0, 1, 2, 0), (1, 0, 2, 0), (1, 1, 2, 1),
    (0, 0, 10, 0), (0, 1, 10, 0), (1, 0, 10, 0), (1, 1, 10, 1), (7, 7, 10, 9), (6, 7, 10, 2), (7, 6, 10, 2),
    (0, 0, M62, 0), (0, 1, M62, 0), (1, 0, M62, 0), (1, 1, M62, 1),
    (2305843009213644800, 2305843009213644801, M62, 1152921504606822400), (2305843009213644801, 2305843009213644800, M62, 1152921504606822400),
    (2305843009213644801, 2305843009213644801, M62, 3458764513820467201), (4611686018427289600, 4611686018427289600, M62, 1),
]


def test_barrett_reduce_128_known_answers():
    L = _lib()
    for lo, hi, m, want in BARRETT:
        assert L.orc_barrett_reduce_128(lo, hi, m) == want == ((hi << 64) | lo) % m, (lo, hi, m)


def test_multiply_uint_uint_mod_known_answers():
    L = _lib()
    for a, b, m, want in MULMOD:
        assert L.orc_mulmod(a, b, m) == want == a * b % m, (a, b, m)
