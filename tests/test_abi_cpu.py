"""CPU-only checks of the product library: it loads, exports every symbol of include/crcnn_hip.h, and its HOST-side
pieces (context tables, encoder, client-side BFV, HDF5 reader) agree with the reference-pinned goldens.
No kernel is launched here (device=-1 contexts); GPU parity lives in tests/test_gpu_*.py."""
import glob
import os

import numpy as np
import pytest

import crcnn_amd as ca
from crcnn_amd import binding
from oracle import orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SETS = sorted(glob.glob(os.path.join(GOLD, "ops_*.npz")))


def test_library_exports_every_declared_symbol():
    L = binding.load()
    syms = binding.header_symbols()
    assert len(syms) > 40
    for s in syms:
        assert hasattr(L, s), s


def test_default_moduli():
    assert ca.default_coeff_modulus_128(4096) == [0x7fffffff380001, 0x3fffffff000001]
    assert ca.default_coeff_modulus_128(8192)[:3] == [0x7fffffff380001, 0x7ffffffef00001, 0x3fffffff000001]
    assert len(ca.default_coeff_modulus_128(16384)) == 8


def test_invalid_parameters_are_rejected():
    with pytest.raises(ca.CrcError):
        ca.Engine(4096, [0x7fffffff380001 + 2], 1 << 20, device=-1)       # not prime / not 1 mod 2n
    with pytest.raises(ca.CrcError):
        ca.Engine(4095, [0x7fffffff380001], 1 << 20, device=-1)
    with pytest.raises(ca.CrcError):
        ca.Engine(4096, [0x7fffffff380001, 0x7fffffff380001], 1 << 20, device=-1)
    E = ca.Engine(4096, [0x7fffffff380001], 1 << 20, device=-1)
    with pytest.raises(ca.CrcError):                                      # device entry point on a host-only context
        E.ntt_fwd(0, 1)


@pytest.mark.parametrize("path", SETS, ids=[os.path.basename(s)[:-4] for s in SETS])
def test_host_tables_and_encoder_match_reference(path):
    g = dict(np.load(path))
    E = ca.Engine(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]), device=-1)
    k = E.k; c = g["ref_consts"]
    assert np.array_equal(E.table("root"), c[:k])
    assert np.array_equal(E.table("const_ratio"), c[k:3 * k])
    assert np.array_equal(E.table("delta"), c[3 * k:4 * k])
    assert np.array_equal(E.table("upper_half_increment"), c[4 * k:5 * k])
    kb = int(c[5 * k]); assert kb == E.kbsk
    assert np.array_equal(E.table("bsk"), c[5 * k + 1:5 * k + 1 + kb])
    assert np.array_equal(E.table("bsk_root"), c[5 * k + 1 + kb:5 * k + 1 + 2 * kb])
    assert np.array_equal(E.table("root_powers:0"), g["ref_root_powers0"][0])
    assert np.array_equal(E.table("inv_root_powers_div_two:0"), g["ref_root_powers0"][1])
    enc, cc = E.encode(g["floats"], dtype=np.float64)
    assert np.array_equal(enc, g["ref_enc_floats"]) and np.array_equal(cc, g["ref_enc_cc"].astype(np.int32))
    for i in range(len(enc)):
        assert E.decode(enc[i]) == g["ref_decode"][i]


@pytest.mark.parametrize("path", SETS, ids=[os.path.basename(s)[:-4] for s in SETS])
def test_compact_plaintexts_are_the_dense_ones(path):
    """crc_encode_f32_compact (96 words per weight: coefficients 0..63 and n-32..n-1, what the host ships to the device) against crc_encode_f32, itself pinned
    to the reference's fraencoder above: the golden's floats, integers of both signs up to 2^62, tiny fractions, and random weights"""
    import ctypes
    g = dict(np.load(path))
    E = ca.Engine(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]), device=-1)
    rng = np.random.default_rng(7)
    vals = np.concatenate([np.asarray(g["floats"], dtype=np.float32), np.array([0, 1, -1, 2.5, -2.5, 3 ** 20, -(3.0 ** 20), 2.0 ** 62, -(2.0 ** 62), 1e-9, -1e-9, 1 / 3, -1 / 3], dtype=np.float32),
                           rng.standard_normal(3000).astype(np.float32), (rng.standard_normal(200) * 1e6).astype(np.float32)])
    dense, cc = E.encode(vals)
    cp = np.zeros((vals.size, E.COMPACT_WORDS), dtype=np.uint64); cc2 = np.zeros(vals.size, dtype=np.int32)
    assert E.L.crc_encode_f32_compact(E.c, vals.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), vals.size, cp.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                      cc2.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))) == 0
    assert np.array_equal(cc, cc2)
    assert np.array_equal(dense[:, :64], cp[:, :64]) and np.array_equal(dense[:, E.n - 32:], cp[:, 64:])
    assert not dense[:, 64:E.n - 32].any()
    E.close()


@pytest.mark.parametrize("path", SETS, ids=[os.path.basename(s)[:-4] for s in SETS])
def test_client_side_interoperates_with_reference(path):
    """engine decrypts the reference's (SEAL-made) ciphertexts; oracle (pinned to SEAL) decrypts the engine's"""
    g = dict(np.load(path))
    n, q, t = int(g["n"]), [int(x) for x in g["q"]], int(g["t"])
    E = ca.Engine(n, q, t, device=-1); O = orc.Oracle(n, q, t)
    assert np.array_equal(E.decrypt(g["ref_sk"], g["ref_enc2"]), g["plains"])
    assert np.array_equal(E.decrypt(g["sk"], g["ref_relin"]), g["ref_dec_relin"])
    assert np.array_equal(E.decrypt(g["sk"], g["ref_sq"], size=3), g["ref_dec_relin"])
    for i in range(len(g["ct_in"])):
        assert E.noise_budget(g["sk"], g["ct_in"][i]) == int(g["ref_budget_in"][i])
        assert E.noise_budget(g["sk"], g["ref_relin"][i]) == int(g["ref_budget_relin"][i])
    sk, pk = E.keygen(77)
    cts = E.encrypt(pk, g["plains"], 5)
    for j in range(len(cts)):
        assert np.array_equal(O.decrypt(sk, cts[j]), g["plains"][j])
        assert O.noise_budget(sk, cts[j]) >= int(g["ref_budget_in"][0]) - 2
    # evaluation keys made by the engine relinearise correctly under the oracle
    evk = E.gen_evk(78, sk)
    r = O.relinearize(O.square(cts[0]), evk)
    if O.noise_budget(sk, r) >= 10:
        v = O.decode(g["plains"][0])
        assert abs(O.decrypt_value(sk, r) - v * v) < 1e-4


def test_seal_layout_roundtrip():
    E = ca.Engine(256, [0x7fffffff380001, 0x3fffffff000001], 1 << 20, device=-1)
    import ctypes
    ct = np.arange(2 * 2 * 256, dtype=np.uint64).reshape(2, 2, 256)
    seal = np.zeros((2, 2, 257), dtype=np.uint64); back = np.zeros_like(ct)
    PU = ctypes.POINTER(ctypes.c_uint64)
    assert E.L.crc_export_seal(E.c, ct.ctypes.data_as(PU), 2, seal.ctypes.data_as(PU)) == 0
    assert np.array_equal(seal[..., :256], ct) and not seal[..., 256].any()
    assert E.L.crc_import_seal(E.c, seal.ctypes.data_as(PU), 2, back.ctypes.data_as(PU)) == 0
    assert np.array_equal(back, ct)
    seal[0, 0, 256] = 1
    assert E.L.crc_import_seal(E.c, seal.ctypes.data_as(PU), 2, back.ctypes.data_as(PU)) < 0


def test_kernel_selection_policy_of_the_abi():
    """crc_plan_mac / crc_plan_fold_pool -- the one statement of which multiply-accumulate kernel a conv / dense layer runs on, asked by both hosts (netrun.py, the C++
    classes) -- on host-only contexts: the layers of the three CrCNN topologies (cnnBuilder.cpp:109-169) at the bench's chunk sizes, and the rules' edges"""
    T = ca.Engine(4096, ca.default_coeff_modulus_128(4096), 1 << 32, device=-1)
    A = ca.Engine(8192, ca.default_coeff_modulus_128(8192)[:3], 1 << 42, device=-1)
    W8 = ca.Engine(16384, ca.default_coeff_modulus_128(16384), 1 << 44, device=-1)
    NTTL, NTTL1, NTTP = ca.NTTL, ca.NTTL1, ca.NTTP
    # PlainModelTiny, chunk 128: conv1+pool1 (one channel, 6 x 6 stride 2) on its own matrix-core kernel, conv2+pool2 / fc3 / fc4 on the limb GEMM
    assert T.plan_mac(1, 28, 28, 2, 2, 6, 6, 32, 128) == NTTL1
    assert T.plan_mac(32, 12, 12, 2, 2, 6, 6, 64, 128) == NTTL
    assert T.plan_mac(1024, 1, 1, 1, 1, 1, 1, 512, 128) == NTTL
    assert T.plan_mac(512, 1, 1, 1, 1, 1, 1, 10, 128) == NTTL
    # ApproxPlainModel, chunk 32: conv1+pool1 folded = 7 x 7 stride 2 (one channel), conv2 = 3 x 3 on 20 channels (9 steps of one padded channel block), fc3 800 -> 500,
    # fc4 500 -> 10 with 64 rows on a ring of n k = 24576
    assert A.plan_mac(1, 28, 28, 2, 2, 7, 7, 20, 32) == NTTL1
    assert A.plan_mac(20, 11, 11, 2, 2, 3, 3, 50, 32) == NTTL
    assert A.plan_mac(800, 1, 1, 1, 1, 1, 1, 500, 32) == NTTL
    assert A.plan_mac(500, 1, 1, 1, 1, 1, 1, 10, 32) == NTTL and A.plan_mac(500, 1, 1, 1, 1, 1, 1, 10, 31) == NTTP
    # fewer than 8 reduction steps of 32 channels: vector ALU (packed operand form)
    assert T.plan_mac(32, 12, 12, 2, 2, 2, 2, 64, 128) == NTTP          # 4 taps x 1 channel block
    assert T.plan_mac(224, 1, 1, 1, 1, 1, 1, 64, 128) == NTTP and T.plan_mac(225, 1, 1, 1, 1, 1, 1, 64, 128) == NTTL
    # fewer than 32 rows per launch (a dense layer has 2 per image): vector ALU; below 24 filters the 64-filter tile is mostly padding: 64 rows, and small rings only
    assert T.plan_mac(1024, 1, 1, 1, 1, 1, 1, 512, 15) == NTTP and T.plan_mac(1024, 1, 1, 1, 1, 1, 1, 512, 16) == NTTL
    assert T.plan_mac(512, 1, 1, 1, 1, 1, 1, 10, 16) == NTTP and T.plan_mac(512, 1, 1, 1, 1, 1, 1, 10, 32) == NTTL
    assert W8.plan_mac(500, 1, 1, 1, 1, 1, 1, 10, 128) == NTTP and W8.plan_mac(800, 1, 1, 1, 1, 1, 1, 500, 32) == NTTL
    # one-channel kernel: window <= 8 x 8, <= 32 filters, image width <= 32; otherwise the generic rules
    assert T.plan_mac(1, 28, 28, 1, 1, 9, 9, 32, 128) != NTTL1 and T.plan_mac(1, 28, 28, 2, 2, 6, 6, 33, 128) != NTTL1 and T.plan_mac(1, 40, 40, 2, 2, 6, 6, 32, 128) != NTTL1
    # matrix_cores = False keeps everything on the vector ALU
    assert T.plan_mac(32, 12, 12, 2, 2, 6, 6, 64, 128, matrix_cores=False) == NTTP and T.plan_mac(1, 28, 28, 2, 2, 6, 6, 32, 128, matrix_cores=False) == NTTP
    # folding a 2 x 2 / 2 average pool into the 5 x 5 convolution in front of it (6 x 6 stride 2: 2.8x fewer multiply-adds) pays
    assert T.plan_fold_pool(1, 28, 28, 1, 1, 5, 5, 32, 2, 2, 2, 2) is True
    assert T.plan_fold_pool(32, 12, 12, 1, 1, 5, 5, 64, 2, 2, 2, 2) is True
    assert A.plan_fold_pool(1, 28, 28, 2, 2, 5, 5, 20, 1, 1, 2, 2) is True       # CrCNN's stride-1 pool behind a one-channel convolution: narrowly


def test_device_encryptor_noise_thresholds_are_the_reference_law():
    """kernels_client.hip samples the encryption noise as |e| = #{a : x >= T_a} on a uniform 64-bit x.  T_a / 2^64 must be P(|e| <= a) of the reference's sampler
    (N(0, 3.19^2), redrawn beyond 6 sigma, static_cast<int64_t>: encryptor.cpp:237-240, util/clipnormal.h) -- checked against the closed form in double
    precision and against that algorithm run literally (numpy) on two million draws."""
    import ctypes
    import math
    L = binding.load()
    out = (ctypes.c_uint64 * 19)()
    L.crc_encrypt_dev_noise_thresholds(out)
    T = [int(v) for v in out]
    assert T == sorted(T) and T[0] > 0 and T[-1] < 2 ** 64
    sigma, lim = 3.19, 6 * 3.19
    Phi = lambda x: 0.5 * math.erfc(-x / (sigma * math.sqrt(2)))
    Z = Phi(lim) - Phi(-lim)
    cum = []
    c = 0.0
    for a in range(19):
        c += (Phi(1) - Phi(-1)) / Z if a == 0 else 2 * (Phi(min(a + 1, lim)) - Phi(a)) / Z
        cum.append(c)
        assert abs(T[a] / 2.0 ** 64 - c) < 1e-13, a
    rng = np.random.default_rng(20260505)
    g = rng.normal(0.0, sigma, size=2_200_000)
    g = g[np.abs(g) <= lim][:2_000_000]                 # redrawing = conditioning
    e = np.abs(np.trunc(g).astype(np.int64))
    N = e.size
    probs = [cum[0]] + [cum[a] - cum[a - 1] for a in range(1, 19)] + [1.0 - cum[18]]
    chi, po, pe = 0.0, 0, 0.0
    for a in range(20):
        ex, ob = probs[a] * N, int((e == a).sum())
        if ex >= 20: chi += (ob - ex) ** 2 / ex
        else: po += ob; pe += ex
    if pe > 0: chi += (po - pe) ** 2 / pe
    assert chi < 60, chi                                 # 99.99 % quantile of chi-square with <= 19 degrees of freedom: 51
