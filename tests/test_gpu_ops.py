"""GPU parity, op level: every Evaluator-op entry point of the C ABI against (a) outputs of the reference itself
(tests/golden/ops_*.npz, produced by SEAL 2.3.1 compiled from /root/reference) and (b) the CPU oracle on seeded inputs
at the BASELINE ring sizes.  Bit-exact (integer arithmetic mod q_i)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SETS = sorted(glob.glob(os.path.join(GOLD, "ops_*.npz")))


@pytest.fixture(scope="module", params=SETS, ids=[os.path.basename(s)[:-4] for s in SETS])
def gs(request):
    import crcnn_amd as ca
    g = dict(np.load(request.param))
    E = ca.Engine(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]), device=0)
    yield g, E
    E.close()


def test_ntt_roundtrip_and_golden(gs):
    g, E = gs
    cts = g["ct_in"]
    d = E.upload(cts)
    E.ntt_fwd(d, len(cts))
    assert np.array_equal(E.download(d, cts.shape), g["ref_ct_ntt"])
    E.ntt_inv(d, len(cts))
    assert np.array_equal(E.download(d, cts.shape), cts)


def test_plain_to_ntt(gs):
    g, E = gs
    pl = g["plains"]
    d_p = E.upload(pl); d_o = E.alloc(len(pl) * E.k * E.n * 8)
    E.plain_to_ntt(d_p, len(pl), d_o)
    assert np.array_equal(E.download(d_o, (len(pl), E.k, E.n)), g["ref_plain_ntt"])


def test_compact_plaintexts_expand_on_the_device(gs):
    """crc_plain_expand: the 96-word compact plaintexts the host ships (crc_encode_f32_compact) zero-extended to [count][n] on the device == the dense
    encoding"""
    g, E = gs
    vals = np.concatenate([np.asarray(g["floats"], dtype=np.float32), np.random.default_rng(3).standard_normal(777).astype(np.float32)])
    dense, _ = E.encode(vals)
    d_p = E.alloc(vals.size * E.n * 8); d_c = E.alloc(vals.size * E.COMPACT_WORDS * 8)
    E.L.crc_memset(E.c, E.p(d_p), 0xff, vals.size * E.n * 8, E.stream)
    assert E.encode_to_device(vals, d_p, d_c) == vals.size
    assert np.array_equal(E.download(d_p, dense.shape), dense)


def test_add(gs):
    g, E = gs
    cts = g["ct_in"]; nct = len(cts)
    d_a = E.upload(cts); d_b = E.upload(np.roll(cts, -1, axis=0))
    E.add(d_a, d_b, nct)
    assert np.array_equal(E.download(d_a, cts.shape), g["ref_add"])


def test_plain_ops(gs):
    import crcnn_amd as ca
    g, E = gs
    cts, pl = g["ct_in"], g["plains"]
    nct, npl = len(cts), len(pl)
    d_p = E.upload(pl)
    d_delta = E.alloc(npl * E.k * E.n * 8); d_w = E.alloc(npl * E.k * E.n * 8)
    E.plain_to_delta(d_p, npl, ca.COEFF, d_delta); E.plain_to_ntt(d_p, npl, d_w)
    rep = np.ascontiguousarray(np.repeat(cts[:, None], npl, axis=1))        # [nct][npl] cts; plaintext index = ct % npl -> use group=1 on a transposed view
    # layout trick: order cts as [npl][nct] so that plaintext j is shared by `nct` consecutive cts (group = nct)
    byp = np.ascontiguousarray(np.transpose(rep, (1, 0, 2, 3, 4)))
    for name, fn in (("ref_add_plain", lambda d: E.add_plain(d, d_delta, nct * npl, nct, +1)),
                     ("ref_sub_plain", lambda d: E.add_plain(d, d_delta, nct * npl, nct, -1)),
                     ("ref_mul_plain", lambda d: E.multiply_plain(d, d_w, nct * npl, nct))):
        d = E.upload(byp); fn(d)
        got = np.transpose(E.download(d, byp.shape), (1, 0, 2, 3, 4))
        assert np.array_equal(got, g[name]), name
    # multiply_plain_ntt on NTT-form cts, then transform_from_ntt
    d = E.upload(byp); E.ntt_fwd(d, nct * npl); E.multiply_plain_ntt(d, d_w, nct * npl, nct)
    assert np.array_equal(np.transpose(E.download(d, byp.shape), (1, 0, 2, 3, 4)), g["ref_mul_ntt"])
    E.ntt_inv(d, nct * npl)
    assert np.array_equal(np.transpose(E.download(d, byp.shape), (1, 0, 2, 3, 4)), g["ref_mul"])


def test_square_and_relinearize(gs):
    g, E = gs
    cts = g["ct_in"]; nct = len(cts)
    d_x = E.upload(cts); d_evk = E.upload(g["evk"])
    d_w = E.alloc(E.square_relin_work_bytes(nct)); d_y3 = E.alloc(nct * 3 * E.k * E.n * 8); d_y = E.alloc(cts.nbytes)
    E.square(d_x, nct, d_y3, d_w)
    assert np.array_equal(E.download(d_y3, (nct, 3, E.k, E.n)), g["ref_sq"])
    E.relinearize(d_y3, nct, d_evk, d_y, d_w)
    assert np.array_equal(E.download(d_y, cts.shape), g["ref_relin"])
    d_y2 = E.alloc(cts.nbytes)
    E.square_relin(d_x, nct, d_evk, d_y2, d_w)
    assert np.array_equal(E.download(d_y2, cts.shape), g["ref_relin"])
    assert np.array_equal(E.download(d_x, cts.shape), cts)          # input untouched
    # between NTT-resident neighbours (crc_square_relin_forms): NTT in and/or out, same ciphertexts in the requested form
    import crcnn_amd as ca
    d_xn = E.upload(cts); E.ntt_fwd(d_xn, nct); xn = E.download(d_xn, cts.shape)
    for fin, fout in [(ca.NTT, ca.NTT), (ca.NTT, ca.COEFF), (ca.COEFF, ca.NTT)]:
        E.square_relin(d_xn if fin == ca.NTT else d_x, nct, d_evk, d_y2, d_w, in_form=fin, out_form=fout)
        if fout == ca.NTT:
            E.ntt_inv(d_y2, nct)
        assert np.array_equal(E.download(d_y2, cts.shape), g["ref_relin"]), (fin, fout)
    assert np.array_equal(E.download(d_xn, cts.shape), xn)


def test_square_pool_with_one_key_switch_per_window(gs):
    """crc_square_pool_relin_forms: Square + relinearise + sum pooling with the digit polynomials of a window added BEFORE the key switch (one key switch per
    pooled
    ciphertext).  Must be the ciphertexts of crc_square_relin_forms followed by crc_pool -- themselves pinned to the reference's relinearize / pooling
    goldens above and
    in test_gpu_layers.py -- AND those of the CPU oracle's square -> relinearise -> pool, bit for bit, in every combination of forms, for CrCNN's overlapping
    2 x 2 / 1 window and a decimating 2 x 2 / 2 one"""
    import crcnn_amd as ca
    g, E = gs
    base = g["ct_in"]
    if not E.square_pool_relin_supported(2, 2):
        pytest.skip("the pooled key switch does not hold this ring's integers")
    rng = np.random.default_rng(11)
    qv = np.array(E.q, dtype=np.uint64).reshape(1, 1, E.k, 1)
    for (B, zd, xd, yd, xs, ys, xf, yf) in [(2, 3, 5, 5, 1, 1, 2, 2), (1, 2, 4, 6, 2, 2, 2, 2), (1, 1, 3, 3, 1, 1, 3, 3)]:
        if not E.square_pool_relin_supported(xf, yf):
            continue
        cnt = B * zd * xd * yd
        # distinct ciphertexts: the golden's encryptions, each plus a different encryption of the set (sums of valid ciphertexts are valid ciphertexts)
        idx = rng.integers(0, len(base), size=(cnt, 2))
        cts = np.ascontiguousarray((base[idx[:, 0]] + base[idx[:, 1]]) % qv)
        xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
        out_cnt = B * zd * xo * yo
        d_evk = E.upload(g["evk"])
        d_x = E.upload(cts); d_xn = E.upload(cts); E.ntt_fwd(d_xn, cnt)
        d_w = E.alloc(max(E.square_relin_work_bytes(cnt), E.square_pool_relin_work_bytes(B, zd, xd, yd, xs, ys, xf, yf)))
        d_r = E.alloc(cts.nbytes); d_p = E.alloc(out_cnt * 2 * E.k * E.n * 8); d_f = E.alloc(out_cnt * 2 * E.k * E.n * 8)
        E.square_relin(d_x, cnt, d_evk, d_r, d_w)
        E.pool(d_r, B, zd, xd, yd, xs, ys, xf, yf, None, ca.COEFF, d_p)
        want = E.download(d_p, (out_cnt, 2, E.k, E.n))
        if E.n <= 4096:
            # ... and of the CPU oracle's Evaluator::square + relinearize followed by PoolingLayer::forward (evaluator.cpp:702-884, 934-1069;
            # poolingLayer.cpp:22-44), image by image: the pooled path is checked against the restatement of the reference, not only against the engine's own
            # unpooled sequence
            from oracle import orc
            O = orc.Oracle(E.n, [int(v) for v in E.q], E.t)
            per = cts.reshape(B, zd, xd, yd, 2, E.k, E.n)
            ow = np.stack([np.asarray(O.pool(O.square_layer(per[b], g["evk"]), xs, ys, xf, yf)) for b in range(B)])
            assert np.array_equal(want, ow.reshape(want.shape)), ("oracle", B, zd, xd, yd, xs, ys, xf, yf)
        for fin, fout in [(ca.COEFF, ca.COEFF), (ca.NTT, ca.NTT), (ca.NTT, ca.COEFF), (ca.COEFF, ca.NTT)]:
            E.L.crc_memset(E.c, E.p(d_f), 0xff, out_cnt * 2 * E.k * E.n * 8, E.stream)
            E.square_pool_relin(d_xn if fin == ca.NTT else d_x, B, zd, xd, yd, xs, ys, xf, yf, d_evk, d_f, d_w, in_form=fin, out_form=fout)
            if fout == ca.NTT:
                E.ntt_inv(d_f, out_cnt)
            assert np.array_equal(E.download(d_f, want.shape), want), (B, zd, xd, yd, xs, ys, xf, yf, fin, fout)
        # an average pooling: the divisor (an NTT-form plaintext) multiplies the pooled tensor inside the last kernel; crc_pool's own product is the reference
        pl, _ = E.encode(np.array([1.0 / (xf * yf)], dtype=np.float32))
        d_pl = E.upload(pl); d_div = E.alloc(E.k * E.n * 8); E.plain_to_ntt(d_pl, 1, d_div)
        E.ntt_fwd(d_r, cnt)
        E.pool(d_r, B, zd, xd, yd, xs, ys, xf, yf, d_div, ca.NTT, d_p)
        E.square_pool_relin(d_xn, B, zd, xd, yd, xs, ys, xf, yf, d_evk, d_f, d_w, in_form=ca.NTT, out_form=ca.NTT, d_div=d_div)
        assert np.array_equal(E.download(d_f, want.shape), E.download(d_p, want.shape)), ("average", xf, yf)
        assert np.array_equal(E.download(d_x, cts.shape), cts)


def test_square_pool_pair_with_all_eight_primes():
    """the pooled key switch at n = 16384 with all eight primes of coeff_modulus_128(16384) -- D = 32 digit polynomials, a 2 x 2 window: the largest integers
    the two
    fp64 primes are asked to hold (2^91 of p_0 p_1 / 2 = 2^92.98) -- against square -> relinearise -> pool one after the other"""
    import crcnn_amd as ca
    n = 16384
    q = ca.default_coeff_modulus_128(n)
    assert len(q) == 8
    E = ca.Engine(n, q, 1 << 44, device=0)
    assert E.square_pool_relin_supported(2, 2)
    sk, pk = E.keygen(5); d_evk = E.upload(E.gen_evk(6, sk))
    pl, _ = E.encode(np.random.default_rng(2).standard_normal(18).astype(np.float32))
    cts = E.encrypt(pk, pl, 77)                                        # 18 ciphertexts: two channels of 3 x 3
    B, zd, xd, yd, xs, ys, xf, yf = 1, 2, 3, 3, 1, 1, 2, 2
    cnt, ocnt = 18, 8
    d_x = E.upload(cts); E.ntt_fwd(d_x, cnt)
    d_w = E.alloc(max(E.square_relin_work_bytes(cnt), E.square_pool_relin_work_bytes(B, zd, xd, yd, xs, ys, xf, yf)))
    d_r = E.alloc(cts.nbytes); d_p = E.alloc(ocnt * 2 * E.k * n * 8); d_f = E.alloc(ocnt * 2 * E.k * n * 8)
    E.square_relin(d_x, cnt, d_evk, d_r, d_w, in_form=ca.NTT, out_form=ca.NTT)
    E.pool(d_r, B, zd, xd, yd, xs, ys, xf, yf, None, ca.NTT, d_p)
    E.square_pool_relin(d_x, B, zd, xd, yd, xs, ys, xf, yf, d_evk, d_f, d_w, in_form=ca.NTT, out_form=ca.NTT)
    want, got = E.download(d_p, (ocnt, 2, E.k, n)), E.download(d_f, (ocnt, 2, E.k, n))
    assert np.array_equal(got, want)
    E.ntt_inv(d_f, ocnt)
    out = E.download(d_f, (ocnt, 2, E.k, n))
    assert E.noise_budget(sk, out[0]) > 100                            # (and it still decrypts: the sum of four squares of small values)
    E.close()


def test_square_with_seal_made_keys(gs):
    """SEAL's evaluation keys hold lazy (non-canonical) residues: same bits required"""
    g, E = gs
    cts = g["ref_enc2"]; nct = len(cts)
    d_x = E.upload(cts); d_evk = E.upload(g["ref_evk"]); d_w = E.alloc(E.square_relin_work_bytes(nct)); d_y = E.alloc(cts.nbytes)
    E.square_relin(d_x, nct, d_evk, d_y, d_w)
    assert np.array_equal(E.download(d_y, cts.shape), g["ref_relin2"])


BIG = [(4096, [0x7fffffff380001, 0x3fffffff000001], 1 << 20),
       (8192, [0x7fffffff380001, 0x7ffffffef00001, 0x3fffffff000001], 1 << 30),
       (16384, [0x7fffffff380001, 0x7ffffffef00001, 0x7ffffffeac0001, 0x7ffffffe700001], 1 << 30),
       (2048, [0x3fffffff000001], 1 << 18)]


@pytest.mark.parametrize("n,q,t", BIG, ids=[f"n{p[0]}_k{len(p[1])}" for p in BIG])
def test_baseline_ring_sizes_vs_oracle(n, q, t):
    """the BASELINE.json parameter sets (n=4096/k=2, n=8192/k=3, n=16384/k=4): NTT, plain ops, square+relin vs the oracle"""
    import crcnn_amd as ca
    from oracle import orc
    O = orc.Oracle(n, q, t); E = ca.Engine(n, q, t, device=0)
    sk, pk = O.keygen(5); evk = O.gen_evk(6, sk)
    vals = np.array([0.5, -1.25, 2.75], dtype=np.float32)
    cts = O.encrypt_many(pk, O.encode_many(vals), 50)
    want_ntt = np.stack([O.ct_to_ntt(c) for c in cts])
    edge = np.zeros((2, 2, len(q), n), dtype=np.uint64)
    edge[1] = (np.array(q, dtype=np.uint64) - np.uint64(1))[None, :, None]
    want_edge = np.stack([O.ct_to_ntt(c) for c in edge])
    # the row transforms both ways round: the round-4 kernels (ntt_wave 0), the ones with one workgroup barrier per transform (15: every ring size
    # that has them, round 5; 31: with inverse butterflies that halve at every stage instead of scaling once) and the engine's own choice (-1)
    for wave in (0, 15, 31, 47, 63, -1):                   # (bit 5, round 6: n = 2048)
        E.set_tuning("ntt_wave", wave)
        d = E.upload(cts)
        E.ntt_fwd(d, 3)
        assert np.array_equal(E.download(d, cts.shape), want_ntt), ("ntt_wave", wave)
        E.ntt_inv(d, 3)
        assert np.array_equal(E.download(d, cts.shape), cts), ("ntt_wave", wave)
        # edge values: all-zero and all-(q-1) polynomials survive the lazy butterflies
        de = E.upload(edge); E.ntt_fwd(de, 2)
        assert np.array_equal(E.download(de, edge.shape), want_edge), ("ntt_wave", wave)
        E.ntt_inv(de, 2)
        assert np.array_equal(E.download(de, edge.shape), edge), ("ntt_wave", wave)
    pl, _ = E.encode(np.array([0.3333, -7.0], dtype=np.float32))
    assert np.array_equal(pl, O.encode_many(np.array([0.3333, -7.0], dtype=np.float32)))
    d_p = E.upload(pl); d_w = E.alloc(2 * len(q) * n * 8); d_dl = E.alloc(2 * len(q) * n * 8)
    E.plain_to_ntt(d_p, 2, d_w); E.plain_to_delta(d_p, 2, ca.COEFF, d_dl)
    assert np.array_equal(E.download(d_w, (2, len(q), n)), O.plains_to_ntt(pl))
    d2 = E.upload(cts[:2]); E.add_plain(d2, d_dl, 2, 1, -1); E.multiply_plain(d2, d_w, 2, 1)
    want = np.stack([O.multiply_plain(O.sub_plain(cts[i], pl[i]), pl[i]) for i in range(2)])
    assert np.array_equal(E.download(d2, want.shape), want)
    # one plaintext per GROUP of ciphertexts (group 2: ciphertexts 0, 1 take plaintext 0, ciphertext 2 plaintext 1), with the dyadic product inside the forward
    # transform's last loop (round 5, the rings that have the wave-local kernel) and as a pass of its own (ntt_wave 0): the oracle's products both ways
    want3 = np.stack([O.multiply_plain(cts[i], pl[i // 2]) for i in range(3)])
    for wave in (0, -1):
        E.set_tuning("ntt_wave", wave)
        d3 = E.upload(cts); E.multiply_plain(d3, d_w, 3, 2)
        assert np.array_equal(E.download(d3, want3.shape), want3), ("multiply_plain", wave)
    E.set_tuning("ntt_wave", -1)
    d_evk = E.upload(evk); d_work = E.alloc(E.square_relin_work_bytes(3)); d_y = E.alloc(cts.nbytes)
    E.square_relin(d, 3, d_evk, d_y, d_work)
    got = E.download(d_y, cts.shape)
    want = O.square_layer(cts, evk, threads=3)
    assert np.array_equal(got, want)
    if O.noise_budget(sk, got[1]) >= 10:
        assert abs(O.decrypt_value(sk, got[1]) - 1.5625) < 1e-4
    E.close()


RELIN_SETS = [(256, 1, 1), (1024, 2, 3), (4096, 2, 3), (8192, 4, 2), (16384, 8, 2)]


@pytest.mark.parametrize("n,k,cnt", RELIN_SETS, ids=[f"n{p[0]}_k{p[1]}" for p in RELIN_SETS])
def test_relinearise_over_fp64_primes_equals_reference_arithmetic(n, k, cnt):
    """Evaluator::relinearize (evaluator.cpp:934-1069) two ways on the same size-3 inputs: key switching over the two fp64 primes + CRT (kernels_relin64.hip,
    the
    default) and over the coefficient moduli (the round-2 kernels, which follow the reference transform by transform), plus the CPU oracle.  Inputs are the
    worst
    cases for the integer bound of the fp64 path: random full-range residues and c2 polynomials whose every digit is 0xffff / whose residues are all q - 1,
    under
    random full-range key material (a key is any element of R_q as far as the arithmetic is concerned).  Every form of crc_square_relin_forms as well."""
    import crcnn_amd as ca
    from oracle import orc
    q = ca.default_coeff_modulus_128(n)[:k] if n >= 4096 else [0x7fffffff380001, 0x3fffffff000001][:k]
    t = 1 << 30
    E = ca.Engine(n, q, t, device=0)
    rng = np.random.default_rng(n + k)
    qa = np.array(q, dtype=np.uint64)
    x3 = np.empty((cnt + 2, 3, k, n), dtype=np.uint64)
    for i in range(k):
        x3[:, :, i] = rng.integers(0, q[i], size=(cnt + 2, 3, n), dtype=np.uint64)
    x3[cnt, 2] = (qa - np.uint64(1))[:, None]                        # c2 = q - 1 everywhere
    # ... and the c2 whose premultiplied form c2 (q/q_i)^-1 mod q_i -- what the digits are cut from -- has its three low 16-bit digits at 0xffff in every
    # coefficient
    for i in range(k):
        v = (q[i] - 1) | 0xffffffffffff
        if v >= q[i]:
            v = (((q[i] >> 48) - 1) << 48) | 0xffffffffffff
        qhat = 1
        for l in range(k):
            if l != i:
                qhat = qhat * q[l] % q[i]
        x3[cnt + 1, 2, i] = np.uint64(v * qhat % q[i])
    evk_words = E.L.crc_evk_words(E.c, 16)
    evk = np.empty(evk_words, dtype=np.uint64)
    rows = evk.reshape(-1, k, n)
    for i in range(k):
        rows[:, i] = rng.integers(0, q[i], size=(rows.shape[0], n), dtype=np.uint64)
    O = orc.Oracle(n, q, t) if n <= 4096 else None
    d_x3 = E.upload(x3); d_evk = E.upload(evk)
    d_w = E.alloc(E.square_relin_work_bytes(cnt + 2))
    outs = {}
    for path in (1, 0):
        E.set_tuning("relin_path", path)
        d_y = E.alloc((cnt + 2) * 2 * k * n * 8)
        E.relinearize(d_x3, cnt + 2, d_evk, d_y, d_w)
        outs[path] = E.download(d_y, (cnt + 2, 2, k, n))
    assert np.array_equal(outs[0], outs[1]), "fp64-prime key switching differs from the transforms over the coefficient moduli"
    if O is not None:
        for i in range(cnt + 2):
            assert np.array_equal(outs[0][i], O.relinearize(x3[i], evk)), ("oracle", i)
    # square + relinearise, all form combinations, both paths (valid ciphertext-shaped inputs: any residues)
    x = np.ascontiguousarray(x3[:, :2])
    d_x = E.upload(x)
    res = {}
    for path in (1, 0):
        E.set_tuning("relin_path", path)
        for fin, fout in [(ca.COEFF, ca.COEFF), (ca.NTT, ca.NTT), (ca.COEFF, ca.NTT), (ca.NTT, ca.COEFF)]:
            d_y = E.alloc(x.nbytes)
            E.square_relin(d_x, cnt + 2, d_evk, d_y, d_w, in_form=fin, out_form=fout)
            res[(path, fin, fout)] = E.download(d_y, x.shape)
    for key, v in res.items():
        if key[0] == 0:
            assert np.array_equal(v, res[(1,) + key[1:]]), key
    # the wider LDS passes of the fp64 transforms (tuning variants: 16 / 32 values per thread, their own LDS swizzles)
    E.set_tuning("relin_path", 0)
    for radix in (4, 5):
        E.set_tuning("f64_radix", radix)
        d_y = E.alloc((cnt + 2) * 2 * k * n * 8)
        E.relinearize(d_x3, cnt + 2, d_evk, d_y, d_w)
        assert np.array_equal(E.download(d_y, (cnt + 2, 2, k, n)), outs[1]), ("radix", radix)
        d_y = E.alloc(x.nbytes)
        E.square_relin(d_x, cnt + 2, d_evk, d_y, d_w, in_form=ca.NTT, out_form=ca.NTT)
        assert np.array_equal(E.download(d_y, x.shape), res[(1, ca.NTT, ca.NTT)]), ("radix", radix, "forms")
    E.set_tuning("f64_radix", 0)
    E.close()


SQ64_SETS = [(256, 1, 1 << 16, 3), (1024, 2, 1 << 30, 3), (4096, 2, 1 << 32, 2), (4096, 2, (1 << 41) - 21, 2), (8192, 3, 1 << 42,
             2), (8192, 4, 1 << 42, 2), (16384, 4, 1 << 44, 2), (16384, 8, 1 << 44, 1)]


@pytest.mark.parametrize("n,k,t,cnt", SQ64_SETS, ids=[f"n{p[0]}_k{p[1]}_t{p[2].bit_length()}" for p in SQ64_SETS])
def test_square_over_fp64_auxiliary_base_equals_reference_base(n, k, t, cnt):
    """Evaluator::square (evaluator.cpp:702-884) two ways on the same inputs: with BEHZ's auxiliary base taken from the engine's fp64 primes
    (kernels_square64.hip, the
    default) and with SEAL's own 61-bit base (the round-2 kernels, which follow baseconverter.cpp constant by constant), plus the CPU oracle (SEAL's base) at
    the
    sizes it finishes in seconds.  The plain moduli are the bench's (2^42 at n = 8192, 2^44 at 16384: the size rule for the number of primes is tight there)
    and the
    inputs include the extremes of the integer bound |t P / q| <= 2 n t q: every residue q_i - 1 (largest products), zero, and 1."""
    import crcnn_amd as ca
    from oracle import orc
    q = ca.default_coeff_modulus_128(n)[:k] if n >= 4096 else [0x7fffffff380001, 0x3fffffff000001][:k]
    E = ca.Engine(n, q, t, device=0)
    primes = [int(v) for v in E.table("sq64_primes")]
    assert len(primes) >= 3, "the fp64 auxiliary base is not in use for this parameter set"
    need = 4 * n * t
    for v in q:
        need *= v
    have = 1
    for v in primes:
        have *= v
    assert have >= need, "prod p_j >= 4 n t q: the base holds |floor(t P / q)| with room for fastbconv_sk's correction"
    assert len(primes) == 3 or have // primes[-1] < 2 * need, "no more primes than the size rule needs"
    rng = np.random.default_rng(7 * n + k)
    qa = np.array(q, dtype=np.uint64)
    x = np.empty((cnt + 4, 2, k, n), dtype=np.uint64)
    for i in range(k):
        x[:, :, i] = rng.integers(0, q[i], size=(cnt + 4, 2, n), dtype=np.uint64)
    x[cnt] = (qa - np.uint64(1))[None, :, None]
    x[cnt + 1] = 0
    x[cnt + 2] = 1
    x[cnt + 3, 0] = (qa - np.uint64(1))[:, None]; x[cnt + 3, 1] = (qa >> np.uint64(1))[:, None]
    N = cnt + 4
    d_x = E.upload(x); d_w = E.alloc(E.square_relin_work_bytes(N))
    outs = {}
    for path in (1, 2):
        E.set_tuning("sq_path", path)
        d_y3 = E.alloc(N * 3 * k * n * 8)
        E.square(d_x, N, d_y3, d_w)
        outs[path] = E.download(d_y3, (N, 3, k, n))
    assert np.array_equal(outs[2], outs[1]), "the square over the fp64 auxiliary base differs from the one over SEAL's base"
    if n <= 4096:
        O = orc.Oracle(n, q, t)
        for i in range(N):
            assert np.array_equal(outs[2][i], O.square(x[i])), ("oracle", i)
    for radix in (4, 5):
        E.set_tuning("f64_radix", radix)
        d_y3 = E.alloc(N * 3 * k * n * 8)
        E.square(d_x, N, d_y3, d_w)
        assert np.array_equal(E.download(d_y3, (N, 3, k, n)), outs[1]), ("radix", radix)
    E.set_tuning("f64_radix", 0)
    # (sq_path 2 and the default 0 both take the fp64 base whenever the parameters fit twelve of the engine's primes; 1 keeps SEAL's 61-bit base)
    # ... and through the NTT-resident entry point with relinearisation behind it (input inverse-transformed inside, third polynomial handed over premultiplied)
    evk = np.empty(E.L.crc_evk_words(E.c, 16), dtype=np.uint64)
    rows = evk.reshape(-1, k, n)
    for i in range(k):
        rows[:, i] = rng.integers(0, q[i], size=(rows.shape[0], n), dtype=np.uint64)
    d_evk = E.upload(evk)
    res = {}
    for path in (1, 2):
        E.set_tuning("sq_path", path)
        for fin, fout in [(ca.COEFF, ca.COEFF), (ca.NTT, ca.NTT)]:
            d_y = E.alloc(x.nbytes)
            E.square_relin(d_x, N, d_evk, d_y, d_w, in_form=fin, out_form=fout)
            res[(path, fin, fout)] = E.download(d_y, x.shape)
    for key, v in res.items():
        if key[0] == 2:
            assert np.array_equal(v, res[(1,) + key[1:]]), key
    # round 4: an NTT-resident square lifts inside its forward transforms (sq_fuse 1) or in a kernel of its own (0); left to itself (-1) the engine picks by k.
    # Every choice gives the reference base's ciphertexts, at every radix of the fp64 transforms
    E.set_tuning("sq_path", 2)
    for fuse in (0, 1):
        E.set_tuning("sq_fuse", fuse)
        for radix in (0, 4):
            E.set_tuning("f64_radix", radix)
            d_y = E.alloc(x.nbytes)
            E.square_relin(d_x, N, d_evk, d_y, d_w, in_form=ca.NTT, out_form=ca.NTT)
            assert np.array_equal(E.download(d_y, x.shape), res[(1, ca.NTT, ca.NTT)]), ("sq_fuse", fuse, "radix", radix)
    E.set_tuning("sq_fuse", -1); E.set_tuning("f64_radix", 0)
    # round 5: the fp64 row kernels with one workgroup barrier per transform (wave-local passes; n = 4096 / 8192 / 16384) -- every kernel of the family switched off
    # (0: the round-4 kernels) and on (15: including the lifting forward kernel the default leaves alone): the reference base's ciphertexts both ways
    if n in (4096, 8192, 16384):                          # (n = 4096 since round 6: two cross stages)
        for wave in (0, 15):
            E.set_tuning("f64_wave", wave)
            for fin, fout in [(ca.COEFF, ca.COEFF), (ca.NTT, ca.NTT)]:
                d_y = E.alloc(x.nbytes)
                E.square_relin(d_x, N, d_evk, d_y, d_w, in_form=fin, out_form=fout)
                assert np.array_equal(E.download(d_y, x.shape), res[(1, fin, fout)]), ("f64_wave", wave, fin, fout)
        E.set_tuning("f64_wave", -1)
        # ... and the 64-bit row transforms of the chain (the two inverse transforms with the product / the exact scaling fused in) likewise
        for wave in (0, 15, 31):
            E.set_tuning("ntt_wave", wave)
            d_y = E.alloc(x.nbytes)
            E.square_relin(d_x, N, d_evk, d_y, d_w, in_form=ca.NTT, out_form=ca.NTT)
            assert np.array_equal(E.download(d_y, x.shape), res[(1, ca.NTT, ca.NTT)]), ("ntt_wave", wave)
        E.set_tuning("ntt_wave", -1)
    E.close()


@pytest.mark.parametrize("n,k,t", [(4096, 2, 1 << 20), (8192, 3, 1 << 30), (1024, 2, 1 << 16)])
def test_device_encryptor(n, k, t):
    """Encryptor::encrypt on the device.  The reference samples from std::random_device, so the check is semantic, and
    the checker is the ORACLE (pinned to SEAL's Decryptor by tests/test_oracle_golden.py), not the product's own client code: every
    ciphertext decrypts under the oracle to its plaintext, its oracle-measured noise budget is that of ciphertexts the oracle's own
    Encryptor restatement makes (same sampling laws), and the stream is a function of (seed / key, ciphertext index)."""
    import crcnn_amd as ca
    from oracle import orc
    q = ca.default_coeff_modulus_128(8192)[:k] if n != 4096 else ca.default_coeff_modulus_128(4096)
    E = ca.Engine(n, q, t, device=0)
    O = orc.Oracle(n, q, t)
    sk, pk = E.keygen(11)
    rng = np.random.default_rng(5)
    cnt = 24
    plains = rng.integers(0, t, size=(cnt, n), dtype=np.uint64)
    plains[0] = 0; plains[1] = t - 1
    d_pk = E.upload(pk); d_pl = E.upload(plains)
    d_ct = E.alloc(cnt * 2 * k * n * 8); d_w = E.alloc(E.encrypt_dev_work_bytes(cnt))
    E.encrypt_dev(d_pk, d_pl, cnt, 77, d_ct, d_w)
    ct = E.download(d_ct, (cnt, 2, k, n))
    assert np.array_equal(np.stack([O.decrypt(sk, ct[i]) for i in range(cnt)]), plains)
    ref = O.encrypt_many(pk, plains[:4], 3)
    assert np.array_equal(np.stack([O.decrypt(sk, ref[i]) for i in range(4)]), plains[:4])
    b_dev = [O.noise_budget(sk, ct[i]) for i in range(cnt)]; b_cpu = [O.noise_budget(sk, ref[i]) for i in range(4)]
    assert min(b_dev) >= min(b_cpu) - 2 and max(b_dev) <= max(b_cpu) + 2, (b_dev, b_cpu)
    # sampling laws of the device generator (ChaCha20 streams): c1 - pk1*u = e2 is not observable without u, but c0 + c1 s = Delta m + v with
    # v = e1 + e2 s - e u: centred, small; its size is what the budget above measures.  Key-based entry point: fresh key, distinct streams
    key = E.random_key()
    E.encrypt_dev_key(d_pk, d_pl, cnt, key, 1000, d_ct, d_w)
    ck = E.download(d_ct, (cnt, 2, k, n))
    assert np.array_equal(np.stack([O.decrypt(sk, ck[i]) for i in range(cnt)]), plains)
    E.encrypt_dev_key(d_pk, d_pl, cnt, key, 1001, d_ct, d_w)         # stream_base + 1: ciphertext i now uses the stream ciphertext i+1 used before
    ck2 = E.download(d_ct, (cnt, 2, k, n))
    assert not np.array_equal(ck2[:, 1], ck[:, 1])
    u_same = (ck2[0, 1].astype(object) - ck[1, 1].astype(object))      # same stream => same u and e2 => identical c1 (c1 does not depend on the plaintext)
    assert not np.any(u_same)
    # deterministic per seed, different across seeds and across ciphertexts
    E.encrypt_dev(d_pk, d_pl, cnt, 77, d_ct, d_w)
    assert np.array_equal(E.download(d_ct, (cnt, 2, k, n)), ct)
    E.encrypt_dev(d_pk, d_pl, cnt, 78, d_ct, d_w)
    assert not np.array_equal(E.download(d_ct, (cnt, 2, k, n))[:, 1], ct[:, 1])
    assert not np.array_equal(ct[2, 1], ct[3, 1])
    # round 5: NTT-form result (three forward transforms per modulus, no inverse one) == crc_ntt_fwd of the coefficient-form result of the same streams
    d_cn = E.alloc(cnt * 2 * k * n * 8)
    E.encrypt_dev_forms(d_pk, d_pl, cnt, 77, ca.NTT, d_cn, d_w)
    d_cc = E.upload(ct); E.ntt_fwd(d_cc, cnt)
    assert np.array_equal(E.download(d_cn, (cnt, 2, k, n)), E.download(d_cc, (cnt, 2, k, n)))
    E.encrypt_dev_key_forms(d_pk, d_pl, cnt, key, 1000, ca.NTT, d_cn, d_w)
    d_cc = E.upload(ck); E.ntt_fwd(d_cc, cnt)
    assert np.array_equal(E.download(d_cn, (cnt, 2, k, n)), E.download(d_cc, (cnt, 2, k, n)))
    E.encrypt_dev_forms(d_pk, d_pl, cnt, 77, ca.COEFF, d_cn, d_w)
    assert np.array_equal(E.download(d_cn, (cnt, 2, k, n)), ct)
    E.close()


def test_device_encryptor_noise_law():
    """The device encryptor draws its noise integers directly from their law (kernels_client.hip: thresholds on a uniform 64-bit word).  Under an all-zero public
    key c0 = e1 + Delta m and c1 = e2 are the noise polynomials themselves: their histogram over 2 x 64 x 4096 draws against the law of the reference's
    sampler -- N(0, 3.19^2), redrawn beyond 6 sigma, truncated toward zero (encryptor.cpp:237-240, util/clipnormal.h) -- computed here independently."""
    import math
    import crcnn_amd as ca
    n, k, t = 4096, 2, 1 << 20
    q = ca.default_coeff_modulus_128(4096)
    E = ca.Engine(n, q, t, device=0)
    cnt = 64
    d_pk = E.upload(np.zeros((2, k, n), dtype=np.uint64)); d_pl = E.upload(np.zeros((cnt, n), dtype=np.uint64))
    d_ct = E.alloc(cnt * 2 * k * n * 8); d_w = E.alloc(E.encrypt_dev_work_bytes(cnt))
    E.encrypt_dev(d_pk, d_pl, cnt, 4242, d_ct, d_w)
    ct = E.download(d_ct, (cnt, 2, k, n))
    q0 = int(q[0])
    e = ct[:, :, 0, :].astype(np.int64); e = np.where(e > q0 // 2, e - q0, e)
    # the same integers under every modulus
    e1 = ct[:, :, 1, :].astype(np.int64); e1 = np.where(e1 > int(q[1]) // 2, e1 - int(q[1]), e1)
    assert np.array_equal(e, e1)
    assert e.min() >= -19 and e.max() <= 19
    sigma, lim = 3.19, 6 * 3.19
    Phi = lambda x: 0.5 * math.erfc(-x / (sigma * math.sqrt(2)))
    Z = Phi(lim) - Phi(-lim)
    law = {}
    for a in range(-19, 20):
        lo, hi = (-1.0, 1.0) if a == 0 else ((a, min(a + 1, lim)) if a > 0 else (max(a - 1, -lim), a))
        law[a] = (Phi(hi) - Phi(lo)) / Z
    assert abs(sum(law.values()) - 1.0) < 1e-12
    N = e.size
    counts = {a: int((e == a).sum()) for a in range(-19, 20)}
    # chi-square over the cells that expect at least 20 draws (the tail beyond is pooled): 99.99 % quantile of chi2 with <= 38 degrees of freedom is below 80
    chi, pooled_obs, pooled_exp, cells = 0.0, 0, 0.0, 0
    for a in range(-19, 20):
        ex = law[a] * N
        if ex >= 20: chi += (counts[a] - ex) ** 2 / ex; cells += 1
        else: pooled_obs += counts[a]; pooled_exp += ex
    if pooled_exp > 0: chi += (pooled_obs - pooled_exp) ** 2 / pooled_exp
    assert chi < 80, (chi, cells, counts)
    assert abs(float(e.mean())) < 5 * sigma / math.sqrt(N) and abs(float(e.std()) - math.sqrt(sum(a * a * p for a, p in law.items()))) < 0.02
    # the thresholds the kernel uses are this law's cumulative probabilities
    T = E.encrypt_dev_noise_thresholds()
    cum = 0.0
    for a in range(19):
        cum += law[a] if a == 0 else 2 * law[a]
        assert abs(T[a] / 2.0 ** 64 - cum) < 1e-13, (a, T[a] / 2.0 ** 64, cum)
    E.close()


def test_device_decryptor_matches_seal(gs):
    """crc_decrypt_dev against SEAL's own Decryptor::decrypt outputs (ref_dec_in of the input ciphertexts, ref_dec_relin of SEAL's relinearised squares), from
    coefficient form and from NTT form, and for SEAL's size-3 squares against the host decryptor (itself pinned by tests/test_abi_cpu.py and the oracle)."""
    import crcnn_amd as ca
    g, E = gs
    n, k = E.n, E.k
    # (ref_relin2: squares SEAL encrypted, squared and relinearised under keys its OWN KeyGenerator made -- ref_sk)
    for cts, want, key in ((g["ct_in"], g["ref_dec_in"], "sk"), (g["ref_relin"], g.get("ref_dec_relin"), "sk"), (g["ref_relin2"], g["ref_dec_relin2"], "ref_sk"),
                           (g["ref_sq"], None, "sk"), (g["ref_sq2"], None, "ref_sk")):
        cnt, size = cts.shape[0], cts.shape[1]
        d_sk = E.upload(g[key])
        if want is None:
            want = E.decrypt(g[key], cts, size=size)
        d_pl = E.alloc(cnt * n * 8)
        d_ct = E.upload(cts)
        d_w = E.alloc(E.decrypt_dev_work_bytes(cnt, size, ca.COEFF))
        E.decrypt_dev(d_sk, d_ct, cnt, d_pl, d_w, size=size, in_form=ca.COEFF)
        assert np.array_equal(E.download(d_pl, (cnt, n)), want)
        assert np.array_equal(E.download(d_ct, cts.shape), cts)            # the input is not modified
        E.ntt_fwd(d_ct, cnt, size=size)
        E.L.crc_memset(E.c, E.p(d_pl), 0xff, cnt * n * 8, E.stream)
        d_w2 = E.alloc(E.decrypt_dev_work_bytes(cnt, size, ca.NTT))
        E.decrypt_dev(d_sk, d_ct, cnt, d_pl, d_w2, size=size, in_form=ca.NTT)
        assert np.array_equal(E.download(d_pl, (cnt, n)), want)


def test_device_fractional_codec(gs):
    """crc_decode_dev / crc_encode_dev_f32 / _f64 == the host encoder (pinned to SEAL's FractionalEncoder by ref_enc_floats / ref_decode in
    tests/test_oracle_golden.py and tests/test_abi_cpu.py), bit for bit, on the golden floats, edge values and 20 000 random ones; and on SEAL's own vectors
    where the golden set carries them."""
    g, E = gs
    n = E.n
    if n <= 96:
        pytest.skip("the fractional encoder needs n > 96")
    rng = np.random.default_rng(17)
    edge = np.array([0.0, -0.0, 0.5, -0.5, 1.5, -1.5, 2.5, 1 / 3, -1 / 3, 1e-9, -1e-9, 12345.678, -98765.4321, 3.0 ** -32, 0.49999997, 1e6 + 0.5, -1e6 - 0.25],
                    dtype=np.float64)
    vals64 = np.concatenate([np.asarray(g["floats"], dtype=np.float64), edge, rng.standard_normal(20000) * 10.0 ** rng.integers(-6, 5, 20000)])
    vals32 = vals64.astype(np.float32)
    for vals, f64 in ((vals32, False), (vals64, True)):
        want, _ = E.encode(vals, dtype=np.float64 if f64 else np.float32)
        d_v = E.upload(vals); d_p = E.alloc(vals.size * n * 8)
        E.L.crc_memset(E.c, E.p(d_p), 0xff, vals.size * n * 8, E.stream)
        E.encode_dev(d_v, vals.size, d_p, f64=f64)
        got = E.download(d_p, (vals.size, n))
        assert np.array_equal(got, want)
        d_o = E.alloc(vals.size * 8)
        E.decode_dev(d_p, vals.size, d_o)
        dec = E.download(d_o, (vals.size,), dtype=np.float64)
        host = np.array([E.decode(want[i]) for i in range(0, vals.size, 97)])
        assert np.array_equal(dec[::97].view(np.uint64), host.view(np.uint64))
    # arbitrary plaintexts (what a decryption at an exhausted budget hands the decoder): the same doubles as the host loop
    junk = rng.integers(0, int(g["t"]), size=(64, n), dtype=np.uint64)
    d_j = E.upload(junk); d_o = E.alloc(64 * 8)
    E.decode_dev(d_j, 64, d_o)
    assert np.array_equal(E.download(d_o, (64,), dtype=np.float64).view(np.uint64), np.array([E.decode(junk[i]) for i in range(64)]).view(np.uint64))
    if "ref_enc_floats" in g:
        fl = np.asarray(g["floats"], dtype=np.float64)
        d_v = E.upload(fl); d_p = E.alloc(fl.size * n * 8)
        E.encode_dev(d_v, fl.size, d_p, f64=True)
        assert np.array_equal(E.download(d_p, (fl.size, n)), g["ref_enc_floats"])
        d_o = E.alloc(fl.size * 8)
        E.decode_dev(d_p, fl.size, d_o)
        assert np.array_equal(E.download(d_o, (fl.size,), dtype=np.float64).view(np.uint64), np.asarray(g["ref_decode"], dtype=np.float64).view(np.uint64))


@pytest.mark.parametrize("n,k,t", [(4096, 2, 1 << 29), (2048, 1, 1 << 18), (8192, 3, 1 << 42)])
def test_device_refresh(n, k, t):
    """crc_refresh_dev == the client-side refresh of Network::forward (network.cpp:30-34): the floats it reports are float(decode(decrypt(ct))) of the host
    client, and every refreshed ciphertext decrypts -- under the ORACLE -- to encode(that float) with a fresh ciphertext's noise budget; all four combinations
    of forms; in place; deterministic per seed and different across seeds."""
    import crcnn_amd as ca
    from oracle import orc
    q = {2048: ca.default_coeff_modulus_128(2048), 4096: ca.default_coeff_modulus_128(4096), 8192: ca.default_coeff_modulus_128(8192)[:3]}[n]
    E = ca.Engine(n, q, t, device=0)
    O = orc.Oracle(n, q, t)
    sk, pk = E.keygen(31)
    rng = np.random.default_rng(9)
    cnt = 40
    vals = (rng.standard_normal(cnt) * 3).astype(np.float32)
    pl, _ = E.encode(vals)
    ct = E.encrypt(pk, pl, 5)
    # give the inputs the shape of mid-network ciphertexts: products of two encodings (fraction digits spread over both ends of the polynomial)
    w, _ = E.encode(np.float32([0.37]))
    d_ct = E.upload(ct); d_w = E.alloc(E.k * n * 8); E.plain_to_ntt(E.upload(w), 1, d_w)
    E.ntt_fwd(d_ct, cnt); E.multiply_plain_ntt(d_ct, d_w, cnt, cnt)
    E.ntt_inv(d_ct, cnt)
    ct = E.download(d_ct, (cnt, 2, k, n))
    dec = E.decrypt(sk, ct)
    want_vals = np.array([np.float32(E.decode(dec[i])) for i in range(cnt)], dtype=np.float32)
    want_plain, _ = E.encode(want_vals)
    d_sk, d_pk = E.upload(sk), E.upload(pk)
    fresh_budget = O.noise_budget(sk, O.encrypt_many(pk, want_plain[:1], 3)[0])
    outs = {}
    for in_form in (ca.COEFF, ca.NTT):
        d_in = E.upload(ct)
        if in_form == ca.NTT: E.ntt_fwd(d_in, cnt)
        for out_form in (ca.COEFF, ca.NTT):
            d_out = E.alloc(cnt * 2 * k * n * 8); d_v = E.alloc(cnt * 4)
            d_work = E.alloc(E.refresh_dev_work_bytes(cnt, in_form))
            E.refresh_dev(d_sk, d_pk, d_in, cnt, 123, d_out, d_work, in_form=in_form, out_form=out_form, d_values=d_v)
            got_vals = E.download(d_v, (cnt,), dtype=np.float32)
            assert np.array_equal(got_vals.view(np.uint32), want_vals.view(np.uint32))
            if out_form == ca.NTT: E.ntt_inv(d_out, cnt)
            r = E.download(d_out, (cnt, 2, k, n))
            assert np.array_equal(np.stack([O.decrypt(sk, r[i]) for i in range(cnt)]), want_plain)
            assert min(O.noise_budget(sk, r[i]) for i in range(0, cnt, 7)) >= fresh_budget - 2
            outs[(in_form, out_form)] = r
    # the re-encryption's randomness is a function of the seed alone: same ciphertexts whatever the forms
    assert all(np.array_equal(v, outs[(ca.COEFF, ca.COEFF)]) for v in outs.values())
    d_in = E.upload(ct); d_work = E.alloc(E.refresh_dev_work_bytes(cnt, ca.COEFF))
    E.refresh_dev(d_sk, d_pk, d_in, cnt, 124, d_in, d_work)                 # in place, another seed, no values wanted
    r2 = E.download(d_in, (cnt, 2, k, n))
    assert not np.array_equal(r2[:, 1], outs[(ca.COEFF, ca.COEFF)][:, 1])
    assert np.array_equal(np.stack([O.decrypt(sk, r2[i]) for i in range(cnt)]), want_plain)
    key = E.random_key()
    E.refresh_dev(d_sk, d_pk, E.upload(ct), cnt, 0, d_in, d_work, key=key, stream_base=77)
    r3 = E.download(d_in, (cnt, 2, k, n))
    assert np.array_equal(np.stack([O.decrypt(sk, r3[i]) for i in range(cnt)]), want_plain)
    E.close()
