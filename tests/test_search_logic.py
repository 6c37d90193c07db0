"""Host logic of the plain-modulus search: the C++ search of crcnn_amd/host/plain_modulus_search.cpp must test
the same moduli in the same order and return the same modulus as the restatement of the reference's recursion
(oracle/search_ref.py <- CrCNN/src/optimalParametersChooser.cpp:30-180) for synthetic predicates.  No GPU work."""
import os
import subprocess

import pytest

from oracle import search_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "crcnn_amd", "lib", "test_host")
Q1 = 18014398492704769          # smaller prime of coeff_modulus_128(4096)


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
    tried = [(int(l.split()[1]), l.split()[2]) for l in out[1:] if l.startswith("tried")]
    return found, tried


def run_ref(lo, hi, first_good, last_good, min_q):
    tried = []
    def pred(t):
        s = ref.MISPREDICTED if t < first_good else ref.OUT_OF_BUDGET if t > last_good else ref.SUCCESS
        tried.append((t, s)); return s
    return ref.search(pred, lo, hi, min_q), tried


CASES = []
for lo_e, hi_e in [(16, 34), (24, 34), (20, 21), (20, 20), (10, 40), (1, 62)]:
    for fg_e in range(lo_e - 1, hi_e + 3, 3):
        for lg_e in (fg_e - 1, fg_e, fg_e + 2, hi_e + 1):
            CASES.append((1 << lo_e, 1 << hi_e, (1 << fg_e) + (fg_e % 2), (1 << max(lg_e, 0)) + 5, Q1))


@pytest.mark.parametrize("case", CASES[::3])
def test_search_matches_reference_control_flow(driver, case):
    assert run_cpp(driver, *case) == run_ref(*case)


def test_second_phase_below_smallest_prime(driver):
    # a power-of-two result >= min q_i triggers the integer search of [2^floor(log2 q), q - 1]  (optimalParametersChooser.cpp:52-63)
    q = (1 << 20) + 7
    for first_good, last_good in [((1 << 20) + 3, 1 << 30), (1 << 20, 1 << 30), ((1 << 21) + 1, 1 << 30), (1 << 22, 1 << 21)]:
        case = (1 << 16, 1 << 34, first_good, last_good, q)
        cpp, py = run_cpp(driver, *case), run_ref(*case)
        assert cpp == py
    found, tried = run_cpp(driver, 1 << 16, 1 << 34, (1 << 20) + 3, 1 << 30, q)
    assert found == (1 << 20) + 3 and any(t % 2 for t, _ in tried)


def test_known_answers(driver):
    # smallest success is returned; out-of-budget everywhere or mispredicted everywhere gives 0
    assert run_cpp(driver, 1 << 16, 1 << 34, 1 << 20, 1 << 28, Q1)[0] == 1 << 20
    assert run_cpp(driver, 1 << 16, 1 << 34, 1 << 40, 1 << 50, Q1)[0] == 0
    assert run_cpp(driver, 1 << 16, 1 << 34, 1, 1 << 10, Q1)[0] == 0
    assert run_cpp(driver, 1 << 24, 1 << 34, 1 << 24, 1 << 34, Q1) == (1 << 24, [(1 << 29, "SUCCESS"), (1 << 26, "SUCCESS"), (1 << 24, "SUCCESS")])
