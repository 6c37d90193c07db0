"""Host logic: the modular-arithmetic header shared by the host table builders and the kernels (crcnn_amd/csrc/modarith.h) against
unsigned __int128 arithmetic -- Barrett (SEAL uintarithsmallmod.h:137-176 semantics), the folding reduction for 2^b - d primes, Shoup
multiplication -- for every prime of the reference's parameter sets.  Compiled with g++ on the spot; no GPU."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("flags", [[], ["-DCRC_FORCE_MAD_MUL"]], ids=["int128", "device-multiply"])
def test_modarith_header_against_int128(flags):
    """-DCRC_FORCE_MAD_MUL compiles the DEVICE form of the 64 x 64 -> 128 multiply (four 32-bit multiply-adds) on the host"""
    exe = os.path.join(tempfile.mkdtemp(), "modarith_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17"] + flags + ["-I", os.path.join(ROOT, "crcnn_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "modarith_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True)
    assert out.startswith("ok "), out
    assert int(out.split()[1]) > 5_000_000


@pytest.mark.parametrize("flags", [[], ["-DCRC_FORCE_MAD_MUL"]], ids=["int128", "device-multiply"])
def test_matrix_core_reduction_against_int128(flags):
    """crcnn_amd/csrc/limbred.h (the once-per-output reduction of kernels_mfma.hip / kernels_mfma1.hip and its bias tables) on the CPU: digit products accumulated in 32-bit
    words from the bias table, reduced, compared with sum x w mod q in 128-bit arithmetic -- random, extreme-digit and edge residues, T = 64 .. 18 000 terms, 40..55-bit moduli"""
    exe = os.path.join(tempfile.mkdtemp(), "limbred_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17"] + flags + ["-I", os.path.join(ROOT, "crcnn_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "limbred_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True)
    assert out.startswith("ok "), out
    assert int(out.split()[1]) > 20_000


def test_fp64_modular_arithmetic_against_int128():
    """crcnn_amd/csrc/f64mod.h (exact modular arithmetic on integers held in doubles: the arithmetic of relinearisation's fp64-prime NTTs) on the CPU: the stated bounds
    on random and extreme operands, the centring reduction at its boundaries, integer -> residue conversion, and a lazy forward / inverse transform pair against an
    O(n^2) evaluation in integers; also pins the primes the engine picks (the largest below 2^47 that are 1 mod 2^16)"""
    exe = os.path.join(tempfile.mkdtemp(), "f64mod_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "crcnn_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "f64mod_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True).split()
    assert out[0] == "ok" and int(out[1]) > 5_000_000
    import crcnn_amd as ca
    E = ca.Engine(4096, [0x7fffffff380001, 0x3fffffff000001], 1 << 20, device=-1)
    assert [int(v) for v in E.table("f64_primes")] == [int(v) for v in out[3:5]]
    for p in E.table("f64_primes"):
        assert int(p) % 65536 == 1 and int(p) < (1 << 47)


def test_square_fp64_auxiliary_base_size_rule():
    """The square's auxiliary base over the engine's fp64 primes (crcnn_amd/csrc/kernels_square64.hip): for every BASELINE parameter set the context takes distinct
    primes below 2^47 that are 1 mod 2^16, the first two being relinearisation's, and just enough of them for prod p_j >= 4 n t q -- BEHZ's fastbconv_sk
    (baseconverter.cpp:448-579) is exact once |floor(t P / q)| / B + #B + 1 < m_sk / 2, with |t P / q| <= 2 n t q for SEAL 2.3.1's non-centred mont_rq"""
    import crcnn_amd as ca
    sets = [(4096, 2, 1 << 32), (8192, 3, 1 << 42), (8192, 4, 1 << 42), (16384, 4, 1 << 44), (16384, 8, 1 << 44), (4096, 2, 1 << 20), (8192, 3, 1 << 30)]
    want = {(4096, 2, 1 << 32): 4, (8192, 3, 1 << 42): 5, (8192, 4, 1 << 42): 6, (16384, 4, 1 << 44): 6, (16384, 8, 1 << 44): 11}
    for n, k, t in sets:
        q = ca.default_coeff_modulus_128(n)[:k]
        E = ca.Engine(n, q, t, device=-1)
        primes = [int(v) for v in E.table("sq64_primes")]
        assert primes[:2] == [int(v) for v in E.table("f64_primes")]
        assert len(set(primes)) == len(primes) >= 3
        need = 4 * n * t
        for v in q:
            need *= v
        have = 1
        for p in primes:
            assert p % 65536 == 1 and p < (1 << 47) and pow(2, p - 1, p) == 1
            have *= p
        assert have >= need + (need >> 27) + (need >> 40)            # the context's rule, exact: room for (1 + k 2^-32)^2 and for the Shenoy-Kumaresan correction
        assert len(primes) == 3 or have // primes[-1] < need + (need >> 27) + (need >> 40), "the fewest primes that satisfy it"
        B, msk = have // primes[-1], primes[-1]
        qq = need // (4 * n * t)
        R_max = (2 * n * t * qq * (2**32 + k) ** 2) // 2**64 + k + 1      # |floor(t P / q)| <= 2 n t q (1 + k 2^-32)^2 + k
        assert R_max // B + len(primes) + 1 < msk // 2
        if (n, k, t) in want:
            assert len(primes) == want[(n, k, t)], (n, k, t, len(primes))
