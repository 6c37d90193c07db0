"""The CPU oracle (oracle/crc_oracle.c) against outputs of the reference itself.

tests/golden/ops_*.npz were produced by oracle/make_golden.py: the oracle's keys/ciphertexts were pushed through the
compiled SEAL 2.3.1 Evaluator (oracle/_ref/ref_harness) and its outputs stored.  Everything here is bit-exact.
"""
import glob
import os

import numpy as np
import pytest

from oracle import orc

SETS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ops_*.npz")))


@pytest.fixture(scope="module", params=SETS, ids=[os.path.basename(s)[:-4] for s in SETS])
def gs(request):
    g = dict(np.load(request.param))
    O = orc.Oracle(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]))
    return g, O


def test_inputs_are_reproducible(gs):
    """the oracle's seeded client side regenerates the stored keys and ciphertexts (RNG drift detector)"""
    g, O = gs
    sk, pk = O.keygen(1000)
    assert np.array_equal(sk, g["sk"]) and np.array_equal(pk, g["pk"])
    assert np.array_equal(O.gen_evk(1001, sk), g["evk"])
    assert np.array_equal(O.encrypt_many(pk, g["msgs"], 2000), g["ct_in"])


def test_context_constants(gs):
    g, O = gs
    k = O.k
    c = g["ref_consts"]
    assert np.array_equal(O.table("root"), c[:k])
    assert np.array_equal(O.table("const_ratio"), c[k:3 * k])
    assert np.array_equal(O.table("delta"), c[3 * k:4 * k])
    assert np.array_equal(O.table("upper_half_increment"), c[4 * k:5 * k])
    kb = int(c[5 * k])
    assert kb == O.kbsk
    assert np.array_equal(O.table("bsk"), c[5 * k + 1:5 * k + 1 + kb])
    assert np.array_equal(O.table("bsk_root"), c[5 * k + 1 + kb:5 * k + 1 + 2 * kb])
    assert np.array_equal(O.table("root_powers:0"), g["ref_root_powers0"][0])
    assert np.array_equal(O.table("inv_root_powers_div_two:0"), g["ref_root_powers0"][1])


def test_encoder(gs):
    g, O = gs
    for i, v in enumerate(g["floats"]):
        co, cc = O.encode(v)
        assert cc == int(g["ref_enc_cc"][i]), v
        assert np.array_equal(co, g["ref_enc_floats"][i]), v
        assert O.decode(co) == g["ref_decode"][i], v
        assert abs(O.decode(co) - v) <= 1e-6 * max(1.0, abs(v))


def test_reference_decrypts_oracle_ciphertexts(gs):
    g, O = gs
    assert np.array_equal(g["ref_dec_in"], g["msgs"])
    for i in range(len(g["ct_in"])):
        assert np.array_equal(O.decrypt(g["sk"], g["ct_in"][i]), g["msgs"][i])
        assert O.noise_budget(g["sk"], g["ct_in"][i]) == int(g["ref_budget_in"][i])


def test_oracle_decrypts_reference_ciphertexts(gs):
    g, O = gs
    for j in range(len(g["plains"])):
        assert np.array_equal(O.decrypt(g["sk"], g["ref_enc"][j]), g["plains"][j])
        assert np.array_equal(O.decrypt(g["ref_sk"], g["ref_enc2"][j]), g["plains"][j])


def test_linear_ops(gs):
    g, O = gs
    nct, npl = len(g["ct_in"]), len(g["plains"])
    for j in range(npl):
        assert np.array_equal(O.plain_to_ntt(g["plains"][j]), g["ref_plain_ntt"][j])
    for i in range(nct):
        ct = g["ct_in"][i]
        ctn = O.ct_to_ntt(ct)
        assert np.array_equal(ctn, g["ref_ct_ntt"][i])
        assert np.array_equal(O.ct_from_ntt(ctn), ct)
        assert np.array_equal(O.add(ct, g["ct_in"][(i + 1) % nct]), g["ref_add"][i])
        for j in range(npl):
            pl = g["plains"][j]
            m = O.multiply_plain_ntt(ctn, g["ref_plain_ntt"][j])
            assert np.array_equal(m, g["ref_mul_ntt"][i, j])
            assert np.array_equal(O.ct_from_ntt(m), g["ref_mul"][i, j])
            assert np.array_equal(O.add_plain(ct, pl), g["ref_add_plain"][i, j])
            assert np.array_equal(O.sub_plain(ct, pl), g["ref_sub_plain"][i, j])
            assert np.array_equal(O.multiply_plain(ct, pl), g["ref_mul_plain"][i, j])


def test_square_relinearize(gs):
    g, O = gs
    for i in range(len(g["ct_in"])):
        s = O.square(g["ct_in"][i])
        assert np.array_equal(s, g["ref_sq"][i])
        r = O.relinearize(s, g["evk"])
        assert np.array_equal(r, g["ref_relin"][i])
        assert np.array_equal(O.decrypt(g["sk"], r), g["ref_dec_relin"][i])
        assert O.noise_budget(g["sk"], r) == int(g["ref_budget_relin"][i])
        if int(g["ref_budget_relin"][i]) >= 10:    # small sets (n=2048,t=2^18: 26-bit fresh budget) are exhausted by one square
            v = O.decode(g["msgs"][i])
            assert abs(O.decrypt_value(g["sk"], r) - v * v) < 1e-4


def test_square_relinearize_with_reference_keys(gs):
    """SEAL-made keys are stored in lazy (non-canonical) NTT form; the oracle must give the same bits with them"""
    g, O = gs
    for j in range(len(g["plains"])):
        s = O.square(g["ref_enc2"][j])
        assert np.array_equal(s, g["ref_sq2"][j])
        r = O.relinearize(s, g["ref_evk"])
        assert np.array_equal(r, g["ref_relin2"][j])
        assert np.array_equal(O.decrypt(g["ref_sk"], r), g["ref_dec_relin2"][j])
