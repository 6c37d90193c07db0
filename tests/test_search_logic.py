"""Host logic of the plain-modulus search, pinned to the REFERENCE.  tests/golden/search_sequences.json holds, for 81 synthetic verdict
tables, the candidates CrCNN's own recursion -- plainModulusBinarySearchInternal, CrCNN/src/optimalParametersChooser.cpp:84-181, compiled in place into
oracle/_ref/search_harness with a table-driven testPlainModulus (oracle/search_harness.cpp, oracle/Makefile) -- tests, in order, and what it returns.
The C++ search of the product (crcnn_amd/host/plain_modulus_search.cpp, driven through test_host searchlogic) must test the same moduli in the same
order and return the same modulus, in both phases: powers of two over [min, max], then -- when the modulus found is not below the smallest coefficient
prime, i.e. SEAL's fast plain lift would be off -- the integers of [2^floor(log2 q), q - 1] (the ten lines of plainModulusBinarySearch, :44-63, that
chain the two calls are what `expected()` below spells out).  No GPU work."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "crcnn_amd", "lib", "test_host")
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "search_sequences.json")))["cases"]
BY_KEY = {(c["lo"], c["hi"], c["pow"], c["first_good"], c["last_good"]): c for c in FIX}
Q1 = 18014398492704769          # smaller prime of coeff_modulus_128(4096)
QS = (1 << 20) + 7              # a small stand-in for it, so that the second phase is reachable with small tables


@pytest.fixture(scope="module")
def driver():
    if not os.path.exists(DRIVER):
        if not os.path.exists(os.path.join(ROOT, "crcnn_amd", "lib", "libcrcnn_hip.so")):
            pytest.fail("libcrcnn_hip.so is missing: run __graft_entry__.build()")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "crcnn_amd", "host")])
    return DRIVER


def run_cpp(driver, lo, hi, first_good, last_good, min_q):
    out = subprocess.check_output([driver, "searchlogic", str(lo), str(hi), str(first_good), str(last_good), str(min_q)], text=True).split("\n")
    found = int(out[0].split()[1])
    tried = [[int(l.split()[1]), l.split()[2]] for l in out[1:] if l.startswith("tried")]
    return found, tried


def expected(lo, hi, first_good, last_good, min_q):
    """optimalParametersChooser.cpp:44-63 over the reference's recorded recursions"""
    p1 = BY_KEY[(lo, hi, 1, first_good, last_good)]
    found, tried = p1["found"], list(p1["tried"])
    if found > 0 and found >= min_q:
        p2 = BY_KEY[(1 << (min_q.bit_length() - 1), min_q - 1, 0, first_good, last_good)]
        tried += p2["tried"]
        if p2["found"] > 0:
            found = p2["found"]
    return found, tried


HAS2 = {(c["first_good"], c["last_good"]) for c in FIX if c["pow"] == 0 and c["lo"] == 1 << 20}
PHASE1 = [c for c in FIX if c["pow"] == 1 and not ((c["lo"], c["hi"]) == (1 << 16, 1 << 34) and (c["first_good"], c["last_good"]) in HAS2)]


@pytest.mark.parametrize("c", PHASE1, ids=[f"{c['lo'].bit_length() - 1}-{c['hi'].bit_length() - 1}-{i}" for i, c in enumerate(PHASE1)])
def test_power_of_two_search_matches_the_reference(driver, c):
    # min_q above every candidate: only the first phase runs
    assert run_cpp(driver, c["lo"], c["hi"], c["first_good"], c["last_good"], 1 << 63) == (c["found"], c["tried"])


TWO = [c for c in FIX if c["pow"] == 1 and (c["lo"], c["hi"]) == (1 << 16, 1 << 34) and (c["first_good"], c["last_good"]) in HAS2]


@pytest.mark.parametrize("c", TWO, ids=[str(i) for i in range(len(TWO))])
def test_second_phase_below_smallest_prime_matches_the_reference(driver, c):
    want = expected(c["lo"], c["hi"], c["first_good"], c["last_good"], QS)
    assert run_cpp(driver, c["lo"], c["hi"], c["first_good"], c["last_good"], QS) == want
    if c["first_good"] == (1 << 20) + 3:            # found 2^21 first, then the smaller non-power 2^20 + 3 with the fast lift on
        assert want[0] == (1 << 20) + 3 and any(t % 2 for t, _ in want[1])


def test_integer_search_over_the_real_prime_gap(driver):
    """[2^53, q1 - 1] for the smaller prime of coeff_modulus_128(4096): 64-bit arithmetic of the non-power recursion"""
    c = BY_KEY[(1 << 53, Q1 - 1, 0, (1 << 53) + 12345, 1 << 60)]
    # drive the product's second phase alone: a first phase over [2^54, 2^54] that returns 2^54 >= q1
    found, tried = run_cpp(driver, 1 << 54, 1 << 54, c["first_good"], c["last_good"], Q1)
    assert tried[0][0] == 1 << 54 and tried[1:] == c["tried"] and found == c["found"]


def test_python_restatement_is_pinned_to_the_reference_too():
    """oracle/search_ref.py (used by tests/test_gpu_host_cpp.py to replay the recursion over verdicts OBSERVED on the GPU, which no fixed table can
    anticipate) reproduces every recorded recursion of the reference"""
    from oracle import search_ref as ref
    for c in FIX:
        tried = []

        def pred(t):
            s = ref.MISPREDICTED if t < c["first_good"] else ref.OUT_OF_BUDGET if t > c["last_good"] else ref.SUCCESS
            tried.append([t, s]); return s
        found = ref.internal(pred, c["lo"], c["hi"], bool(c["pow"]))
        assert (found, tried) == (c["found"], c["tried"]), c
