"""GPU parity, layer level: crc_conv2d / crc_dense / crc_pool / crc_batchnorm / crc_square_relin against the outputs of
the reference's own Layer::forward (tests/golden/layers_*.npz, CrCNN/src/*Layer.cpp compiled in place) and, for batches
and NTT-resident chains, against the CPU oracle.  Bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden", "layers_n256_k2_t20.npz")


class Net:
    """helper: uploads encoded parameters the way the C++ host classes do"""

    def __init__(self, E):
        import crcnn_amd as ca
        self.E, self.ca = E, ca

    def weights_ntt(self, w):
        pl, _ = self.E.encode(np.asarray(w, dtype=np.float32))
        d_p = self.E.upload(pl); d_w = self.E.alloc(len(pl) * self.E.k * self.E.n * 8)
        self.E.plain_to_ntt(d_p, len(pl), d_w)
        return d_w

    def delta(self, v, form, dtype=np.float32):
        pl, _ = self.E.encode(np.asarray(v, dtype=dtype), dtype=dtype)
        d_p = self.E.upload(pl); d_o = self.E.alloc(len(pl) * self.E.k * self.E.n * 8)
        self.E.plain_to_delta(d_p, len(pl), form, d_o)
        return d_o


@pytest.fixture(scope="module")
def gl():
    import crcnn_amd as ca
    g = dict(np.load(G))
    E = ca.Engine(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]), device=0)
    yield g, E, Net(E)
    E.close()


def ctshape(E, *lead):
    return tuple(lead) + (2, E.k, E.n)


def test_conv2d_matches_reference_layer(gl):
    g, E, N = gl
    ca = N.ca
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    d_w = N.weights_ntt(g["conv_w"]); d_b = N.delta(g["conv_b"], ca.COEFF)
    for B in (1, 3):
        x = np.ascontiguousarray(np.repeat(g["x"][None], B, axis=0))
        d_x = E.upload(x); d_y = E.alloc(B * nf * xo * yo * 2 * E.k * E.n * 8)
        d_work = E.alloc(E.conv2d_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF))
        E.conv2d(d_x, d_w, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.COEFF, d_y, d_work)
        y = E.download(d_y, ctshape(E, B, nf, xo, yo))
        for b in range(B):
            assert np.array_equal(y[b], g["ref_conv"])
        assert np.array_equal(E.download(d_x, x.shape), x)


def test_dense_matches_reference_layer(gl):
    g, E, N = gl
    ca = N.ca
    out_dim, in_dim = g["fc_w"].shape
    d_w = N.weights_ntt(g["fc_w"]); d_b = N.delta(g["fc_b"], ca.COEFF)
    d_x = E.upload(g["x"]); d_y = E.alloc(out_dim * 2 * E.k * E.n * 8)
    d_work = E.alloc(E.dense_work_bytes(1, in_dim, out_dim, ca.COEFF))
    E.dense(d_x, d_w, d_b, 1, in_dim, out_dim, ca.COEFF, ca.COEFF, d_y, d_work)
    assert np.array_equal(E.download(d_y, ctshape(E, 1, out_dim, 1)), g["ref_fc"])


def test_pools_match_reference_layers(gl):
    g, E, N = gl
    ca = N.ca
    zd, xd, yd = [int(v) for v in g["dims"][:3]]
    pxs, pys, pxf, pyf = [int(v) for v in g["dims"][9:13]]
    xo, yo = (xd - pxf) // pxs + 1, (yd - pyf) // pys + 1
    d_x = E.upload(g["x"]); d_y = E.alloc(zd * xo * yo * 2 * E.k * E.n * 8)
    E.pool(d_x, 1, zd, xd, yd, pxs, pys, pxf, pyf, None, ca.COEFF, d_y)
    assert np.array_equal(E.download(d_y, ctshape(E, zd, xo, yo)), g["ref_pool"])
    div = N.weights_ntt(np.array([1.0 / (pxf * pyf)]))          # float32(0.25) == double 0.25: encode(1./(xf*yf)), avgPoolingLayer.cpp:12
    E.pool(d_x, 1, zd, xd, yd, pxs, pys, pxf, pyf, div, ca.COEFF, d_y)
    assert np.array_equal(E.download(d_y, ctshape(E, zd, xo, yo)), g["ref_avgpool"])


def test_batchnorm_matches_reference_layer(gl):
    import ctypes
    g, E, N = gl
    ca = N.ca
    zd, xd, yd = [int(v) for v in g["dims"][:3]]
    var = np.ascontiguousarray(g["bn_var"], dtype=np.float32); invstd = np.zeros_like(var)
    FP = ctypes.POINTER(ctypes.c_float)
    assert E.L.crc_bn_invstd_f32(var.ctypes.data_as(FP), var.size, invstd.ctypes.data_as(FP)) == 0
    d_mean = N.delta(g["bn_mean"], ca.COEFF); d_inv = N.weights_ntt(invstd)
    d_x = E.upload(g["x"])
    E.batchnorm(d_x, 1, zd, xd, yd, d_mean, d_inv, ca.COEFF)
    assert np.array_equal(E.download(d_x, g["x"].shape), g["ref_bn"])


def test_square_layer_matches_reference_layer(gl):
    g, E, N = gl
    cnt = int(np.prod(g["x"].shape[:3]))
    d_x = E.upload(g["x"]); d_evk = E.upload(g["evk"]); d_y = E.alloc(g["x"].nbytes); d_w = E.alloc(E.square_relin_work_bytes(cnt))
    E.square_relin(d_x, cnt, d_evk, d_y, d_w)
    assert np.array_equal(E.download(d_y, g["x"].shape), g["ref_square"])


def test_ntt_resident_chain_equals_reference_order(gl):
    """conv -> avgpool -> batchnorm -> dense kept in NTT form end to end, one INTT at the very end, must
    give the bits of the reference's coefficient-form layer sequence (computed here with the oracle's reference-order loops)"""
    from oracle import orc
    g, E, N = gl
    ca = N.ca
    O = orc.Oracle(E.n, [int(v) for v in E.q], E.t)
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1                # 2 x 3
    rng = np.random.RandomState(11)
    mean = rng.normal(0, 0.2, nf).astype(np.float32); var = rng.uniform(0.5, 2, nf).astype(np.float32)
    invstd = np.float32(1.0 / np.sqrt(var.astype(np.float64) + 0.00001))
    pxo, pyo = xo - 1, yo - 1                                        # 2x2 window stride 1
    fcw = rng.normal(0, 0.3, (3, nf * pxo * pyo)).astype(np.float32); fcb = rng.normal(0, 0.1, 3).astype(np.float32)
    # --- oracle, reference order, coefficient form between layers
    enc = lambda a: O.encode_many(np.asarray(a, dtype=np.float32)).reshape(np.shape(a) + (O.n,))
    y = O.conv(g["x"], O.plains_to_ntt(enc(g["conv_w"])), enc(g["conv_b"]), xs, ys)
    y = O.pool(y, 1, 1, 2, 2, div_plain=O.encode(0.25)[0])
    y = O.bn(y, enc(mean), enc(invstd))
    want = O.fc(y, O.plains_to_ntt(enc(fcw)), enc(fcb))
    # --- engine, NTT resident, batch of 2 identical images
    B = 2
    x = np.ascontiguousarray(np.repeat(g["x"][None], B, axis=0))
    d_x = E.upload(x)
    d_y1 = E.alloc(B * nf * xo * yo * 2 * E.k * E.n * 8); d_y2 = E.alloc(B * nf * pxo * pyo * 2 * E.k * E.n * 8); d_y3 = E.alloc(B * 3 * 2 * E.k * E.n * 8)
    d_work = E.alloc(max(E.conv2d_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF), E.dense_work_bytes(B, nf * pxo * pyo, 3, ca.NTT)))
    E.conv2d(d_x, N.weights_ntt(g["conv_w"]), N.delta(g["conv_b"], ca.NTT), B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTT, d_y1, d_work)
    E.pool(d_y1, B, nf, xo, yo, 1, 1, 2, 2, N.weights_ntt(np.array([0.25])), ca.NTT, d_y2)
    E.batchnorm(d_y2, B, nf, pxo, pyo, N.delta(mean, ca.NTT), N.weights_ntt(invstd), ca.NTT)
    E.dense(d_y2, N.weights_ntt(fcw), N.delta(fcb, ca.NTT), B, nf * pxo * pyo, 3, ca.NTT, ca.NTT, d_y3, d_work)
    E.ntt_inv(d_y3, B * 3)
    got = E.download(d_y3, ctshape(E, B, 1, 3, 1))
    for b in range(B):
        assert np.array_equal(got[b], want)
    vals = [O.decrypt_value(g["sk"], got[0, 0, i, 0]) for i in range(3)]
    assert all(abs(v) < 50 for v in vals)


def test_conv_pool_fusion_is_exact(gl):
    """crc_conv2d_fold_pool: pool(conv(x)+b) computed as ONE convolution with the pooled kernel equals the reference's
    conv layer followed by its (avg / sum) pooling layer, bit for bit (both are linear over Z_q)"""
    from oracle import orc
    g, E, N = gl
    ca = N.ca
    O = orc.Oracle(E.n, [int(v) for v in E.q], E.t)
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    for avg in (True, False):
        pxf, pyf, pxs, pys = 2, 2, 1, 1
        want = O.pool(g["ref_conv"], pxs, pys, pxf, pyf, div_plain=O.encode(0.25)[0] if avg else None)
        d_w = N.weights_ntt(g["conv_w"]); d_b = N.delta(g["conv_b"], ca.NTT)
        d_div = N.weights_ntt(np.array([0.25])) if avg else None
        xf2, yf2 = (pxf - 1) * xs + xf, (pyf - 1) * ys + yf
        d_w2 = E.alloc(nf * zd * xf2 * yf2 * E.k * E.n * 8); d_b2 = E.alloc(nf * E.k * E.n * 8)
        E.conv2d_fold_pool(d_w, d_b, d_div, nf, zd, xf, yf, xs, ys, pxf, pyf, d_w2, d_b2)
        pxo, pyo = (xo - pxf) // pxs + 1, (yo - pyf) // pys + 1
        assert ((xd - xf2) // (xs * pxs) + 1, (yd - yf2) // (ys * pys) + 1) == (pxo, pyo)
        B = 2
        x = np.ascontiguousarray(np.repeat(g["x"][None], B, axis=0))
        d_x = E.upload(x); d_y = E.alloc(B * nf * pxo * pyo * 2 * E.k * E.n * 8)
        d_work = E.alloc(E.conv2d_work_bytes(B, zd, xd, yd, xs * pxs, ys * pys, xf2, yf2, nf, ca.COEFF))
        E.conv2d(d_x, d_w2, d_b2, B, zd, xd, yd, xs * pxs, ys * pys, xf2, yf2, nf, ca.COEFF, ca.NTT, d_y, d_work)
        E.ntt_inv(d_y, B * nf * pxo * pyo)
        got = E.download(d_y, ctshape(E, B, nf, pxo, pyo))
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def test_shape_validation(gl):
    g, E, N = gl
    ca = N.ca
    # stride larger than window with a remainder: the reference leaves empty ciphertexts there -> rejected, not invented
    assert E.conv2d_work_bytes(1, 1, 5, 5, 3, 3, 2, 2, 1, ca.COEFF) == 0
    with pytest.raises(ca.CrcError):
        E.pool(1, 1, 1, 5, 5, 3, 3, 2, 2, None, ca.COEFF, 1)
    with pytest.raises(ca.CrcError):
        E.conv2d(1, 1, 1, 1, 1, 2, 2, 1, 1, 3, 3, 1, ca.COEFF, ca.COEFF, 1, 1)     # filter larger than image


RAGGED = [
    # zd, xd, yd, xs, ys, xf, yf, nf, B      (T = zd*xf*yf reduction terms, M = B*xo*yo pixels)
    (1, 5, 5, 1, 1, 3, 3, 1, 1),             # T=9 (odd: half-filled last stage), one filter
    (3, 7, 6, 2, 1, 3, 2, 5, 2),             # T=18, F=5 (one partial filter tile), M=2*3*5=30
    (7, 4, 4, 1, 1, 2, 2, 17, 3),            # T=28, F=17 (12x8 shape: 24, 6x16: 32), M=27
    (11, 3, 3, 1, 1, 3, 3, 20, 1),           # T=99 > 64: overflow parking of the lazy accumulators, M=1
    (70, 1, 1, 1, 1, 1, 1, 50, 7),           # dense-shaped, T=70, F=50, M=7
    (2, 6, 6, 3, 3, 3, 3, 9, 0),             # empty batch
    (12001, 1, 1, 1, 1, 1, 1, 2, 1),         # T=12001: term table no longer fits beside mac3's stage buffers -> register-staged mac2
    (16500, 1, 1, 1, 1, 1, 1, 2, 1),         # T=16500: past the LDS term table altogether -> mac_kernel (v1)
]


@pytest.mark.parametrize("geo", RAGGED, ids=[f"z{g[0]}_{g[1]}x{g[2]}_s{g[3]}{g[4]}_f{g[5]}x{g[6]}_nf{g[7]}_B{g[8]}" for g in RAGGED])
def test_mac_ragged_geometries_exact(geo):
    """crc_conv2d on NTT-form operands is, per slot, y[b,f,i,j] = sum_{z,u,v} x[b,z,i*xs+u,j*ys+v] * w[f,z,u,v] + bias[f] (poly 0)
    mod q_i (convolutionalLayer.cpp:56-93 with multiply_plain_ntt, evaluator.cpp:1541-1585): checked against exact integer
    arithmetic on random residues for shapes that leave partial pixel / filter tiles, odd reduction lengths, sums past
    2^64 per limb, and an empty batch."""
    import crcnn_amd as ca
    zd, xd, yd, xs, ys, xf, yf, nf, B = geo
    n, q = 128, ca.default_coeff_modulus_128(4096)
    E = ca.Engine(n, q, 1 << 20, device=0)
    k = E.k
    rng = np.random.default_rng(abs(hash(geo)) % (1 << 32))
    qa = np.array(q, dtype=np.uint64).reshape(1, k, 1)
    def rnd(*lead):
        hi = rng.integers(0, 1 << 62, size=lead + (k, n), dtype=np.uint64)
        return hi % qa
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    x = rnd(max(B, 1), zd, xd, yd, 2)[:B]; w = rnd(nf, zd, xf, yf); bias = rnd(nf)
    d_x = E.upload(x) if B else E.alloc(64); d_w = E.upload(w); d_b = E.upload(bias)
    d_y = E.alloc(max(B, 1) * nf * xo * yo * 2 * k * n * 8)
    d_work = E.alloc(E.conv2d_work_bytes(max(B, 1), zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT))
    E.conv2d(d_x, d_w, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTT, d_y, d_work)
    E.sync()
    if B == 0:
        E.close(); return
    y = E.download(d_y, (B, nf, xo, yo, 2, k, n))
    xo_ = x.astype(object); wo = w.astype(object); qo = [int(v) for v in q]
    for b in range(B):
        for i in range(xo):
            for j in range(yo):
                patch = xo_[b, :, i * xs:i * xs + xf, j * ys:j * ys + yf]                 # [zd][xf][yf][2][k][n]
                for f in range(nf):
                    acc = (patch * wo[f][:, :, :, None]).sum(axis=(0, 1, 2))               # [2][k][n] exact integers
                    acc[0] = acc[0] + bias[f].astype(object)
                    for m in range(k):
                        exp = np.array([int(v) % qo[m] for v in acc[:, m].reshape(-1)], dtype=np.uint64).reshape(2, n)
                        assert np.array_equal(y[b, f, i, j, :, m], exp), (geo, b, f, i, j, m)
    E.close()


def test_mac_operand_offsets_past_4GiB():
    """an input tensor larger than 4 GiB (40 channels of 32x32 ciphertexts at n=4096, k=2 = 5.4 GB): operand byte offsets no longer fit
    32 bits; spot-check output pixels against exact integer arithmetic.  The input is filled by the device encryptor (any residues do)."""
    import crcnn_amd as ca
    n, q = 4096, ca.default_coeff_modulus_128(4096)
    E = ca.Engine(n, q, 1 << 20, device=0)
    k = E.k
    zd, xd, yd, nf = 40, 32, 32, 2
    ctb = 2 * k * n * 8
    cts = zd * xd * yd
    assert cts * ctb > (1 << 32)
    sk, pk = E.keygen(5)
    d_pk = E.upload(pk)
    chunk = 4096
    d_pl = E.upload(np.zeros((chunk, n), dtype=np.uint64)); d_w8 = E.alloc(E.encrypt_dev_work_bytes(chunk))
    d_x = E.alloc(cts * ctb)
    for o in range(0, cts, chunk):
        E.encrypt_dev(d_pk, d_pl, chunk, 1000 + o, E.p(d_x) + o * ctb, d_w8)
    rng = np.random.default_rng(9)
    qa = np.array(q, dtype=np.uint64).reshape(1, k, 1)
    w = rng.integers(0, 1 << 62, size=(nf * zd, k, n), dtype=np.uint64) % qa
    bias = rng.integers(0, 1 << 62, size=(nf, k, n), dtype=np.uint64) % qa
    d_w = E.upload(w); d_b = E.upload(bias)
    d_y = E.alloc(nf * xd * yd * ctb)
    d_work = E.alloc(E.conv2d_work_bytes(1, zd, xd, yd, 1, 1, 1, 1, nf, ca.NTT))
    E.conv2d(d_x, d_w, d_b, 1, zd, xd, yd, 1, 1, 1, 1, nf, ca.NTT, ca.NTT, d_y, d_work)
    E.sync()
    wv = w.reshape(nf, zd, k, n).astype(object)
    for (i, j) in [(0, 0), (31, 31), (17, 5), (31, 0)]:
        px = np.stack([E.download(E.p(d_x) + ((z * xd + i) * yd + j) * ctb, (2, k, n)) for z in range(zd)]).astype(object)       # [zd][2][k][n]
        for f in range(nf):
            acc = (px * wv[f][:, None]).sum(axis=0)                                          # [2][k][n]
            acc[0] = acc[0] + bias[f].astype(object)
            got = E.download(E.p(d_y) + ((f * xd + i) * yd + j) * ctb, (2, k, n))
            for m in range(k):
                exp = np.array([int(v) % int(q[m]) for v in acc[:, m].reshape(-1)], dtype=np.uint64).reshape(2, n)
                assert np.array_equal(got[:, m], exp), (i, j, f, m)
    E.close()


def test_packed_operand_forms(gl):
    """CRC_NTTP (28-bit limb pairs, the MAC kernels' operand form) in any position -- input, weights, output -- gives the ciphertexts
    of the canonical NTT path; crc_pack28 round-trips"""
    g, E, N = gl
    ca = N.ca
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    d_w = N.weights_ntt(g["conv_w"]); d_b = N.delta(g["conv_b"], ca.NTT)
    B = 3
    x = np.ascontiguousarray(np.repeat(g["x"][None], B, axis=0))
    d_x = E.upload(x); E.ntt_fwd(d_x, B * zd * xd * yd)
    rows_x, rows_w, rows_y = B * zd * xd * yd * 2 * E.k, nf * zd * xf * yf * E.k, B * nf * xo * yo * 2 * E.k
    d_y = E.alloc(rows_y * E.n * 8); d_work = E.alloc(E.conv2d_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF))
    E.conv2d(d_x, d_w, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTT, d_y, d_work)
    want = E.download(d_y, (rows_y, E.n))
    xn = E.download(d_x, (rows_x, E.n)); wn = E.download(d_w, (rows_w, E.n))
    d_xp = E.upload(xn); E.pack28(d_xp, rows_x); d_wp = E.upload(wn); E.pack28(d_wp, rows_w)
    packed = E.download(d_xp, (rows_x, E.n))
    assert np.array_equal(packed & np.uint64(0xffffffff), xn & np.uint64(0x0fffffff)) and np.array_equal(packed >> np.uint64(32), xn >> np.uint64(28))
    for fin, fw, fout in [(ca.NTTP, ca.NTTP, ca.NTTP), (ca.NTT, ca.NTTP, ca.NTT), (ca.NTTP, ca.NTT, ca.NTT), (ca.NTT, ca.NTT, ca.NTTP), (ca.NTTP, ca.NTTP, ca.COEFF)]:
        E.conv2d(d_xp if fin == ca.NTTP else d_x, d_wp if fw == ca.NTTP else d_w, d_b if fout != ca.COEFF else N.delta(g["conv_b"], ca.COEFF), B, zd, xd, yd, xs, ys, xf, yf, nf,
                 fin, fout, d_y, d_work, w_form=fw)
        if fout == ca.NTTP:
            E.pack28(d_y, rows_y, unpack=True)
        if fout == ca.COEFF:
            E.ntt_fwd(d_y, B * nf * xo * yo)
        assert np.array_equal(E.download(d_y, (rows_y, E.n)), want), (fin, fw, fout)
    E.pack28(d_xp, rows_x, unpack=True)
    assert np.array_equal(E.download(d_xp, (rows_x, E.n)), xn)
    # coefficient-form input (the private NTT copy is written packed) still gives the reference layer's ciphertexts
    E.conv2d(E.upload(x), d_wp, N.delta(g["conv_b"], ca.COEFF), B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.COEFF, d_y, d_work, w_form=ca.NTTP)
    y = E.download(d_y, (B, nf, xo, yo, 2, E.k, E.n))
    for b in range(B):
        assert np.array_equal(y[b], g["ref_conv"])


@pytest.mark.parametrize("n,k,in_dim,out_dim", [(1024, 2, 70, 7), (512, 3, 33, 10), (4096, 2, 200, 5)])
def test_dense_at_batch_one_streams_its_weights(n, k, in_dim, out_dim):
    """mac_stream_kernel (a dense layer on ONE image: two rows per weight, the weight stream of the single-image latency line) against exact integer arithmetic on
    random residues -- sums past 2^64 per limb (in_dim > 32), a filter count that is no multiple of the filter group, every shape of the kernel (CRC_MAC_STREAM
    1..4) and mac3_kernel (0), canonical and packed operands in and out"""
    import crcnn_amd as ca
    q = ca.default_coeff_modulus_128(8192)[:k] if k == 3 else ca.default_coeff_modulus_128(4096)
    E = ca.Engine(n, q, 1 << 20, device=0)
    rng = np.random.default_rng(n + in_dim)
    qa = np.array(q, dtype=np.uint64).reshape(1, k, 1)
    x = rng.integers(0, 1 << 62, size=(in_dim, 2, k, n), dtype=np.uint64) % qa.reshape(1, 1, k, 1)
    w = rng.integers(0, 1 << 62, size=(out_dim * in_dim, k, n), dtype=np.uint64) % qa
    x[0] = (qa - 1).reshape(1, k, 1); w[:in_dim] = qa - 1                 # the largest products on filter 0
    bias = rng.integers(0, 1 << 62, size=(out_dim, k, n), dtype=np.uint64) % qa
    qo = [int(v) for v in q]
    xo, wo = x.astype(object), w.reshape(out_dim, in_dim, k, n).astype(object)
    want = np.zeros((out_dim, 2, k, n), dtype=np.uint64)
    for f in range(out_dim):
        acc = (xo * wo[f][:, None]).sum(axis=0)                            # [2][k][n] exact integers
        acc[0] = acc[0] + bias[f].astype(object)
        for m in range(k):
            want[f, :, m] = np.array([int(v) % qo[m] for v in acc[:, m].reshape(-1)], dtype=np.uint64).reshape(2, n)
    d_x = E.upload(x); d_w = E.upload(w); d_b = E.upload(bias)
    d_y = E.alloc(out_dim * 2 * k * n * 8)
    d_work = E.alloc(E.dense_work_bytes(1, in_dim, out_dim, ca.NTT))
    for shape in (1, 2, 3, 4, 5, 0):
        E.set_tuning("mac_stream", shape)
        E.L.crc_memset(E.c, E.p(d_y), 0xff, out_dim * 2 * k * n * 8, E.stream)
        E.dense(d_x, d_w, d_b, 1, in_dim, out_dim, ca.NTT, ca.NTT, d_y, d_work)
        assert np.array_equal(E.download(d_y, want.shape), want), shape
    # packed operands (CRC_NTTP: what Network::forward hands a dense layer that follows another one) in, packed out
    E.set_tuning("mac_stream", 1)
    d_xp = E.upload(x); E.pack28(d_xp, in_dim * 2 * k)
    d_wp = E.upload(w); E.pack28(d_wp, out_dim * in_dim * k)
    E.dense(d_xp, d_wp, d_b, 1, in_dim, out_dim, ca.NTTP, ca.NTTP, d_y, d_work, w_form=ca.NTTP)
    E.pack28(d_y, out_dim * 2 * k, unpack=True)
    assert np.array_equal(E.download(d_y, want.shape), want)
    E.close()
