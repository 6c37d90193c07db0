"""GPU parity, layer level, of the matrix-core kernels (kernels_mfma.hip / kernels_mfma1.hip):
  * DIRECTLY against the CPU oracle's convolution (oracle/crc_oracle.c conv_run = convolution3d, convolutionalLayer.cpp:56-93) on the same operands -- random
    full-range residues and extreme-digit residues (the worst cases for the balanced-limb split: digits at both ends of [-128, 127]) as the NTT-domain values --
    for every weight form (CRC_NTTL, CRC_NTTL1), the coefficient-form layer contract, NTT residency, and the limb hand-overs (CRC_NTTLC, CRC_NTTL), whose
    expected tensors are built here in numpy from the oracle's result;
  * against the reference's own ConvolutionalLayer / FullyConnectedLayer outputs (tests/golden/layers*_n256_k2_t20.npz) with the weights in limb form;
  * against the vector-ALU kernel of the same layer for the remaining operand / result forms, ragged shapes and multi-pass runs.
Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q = [0x7fffffff380001, 0x3fffffff000001]          # 55 and 54 bits: the widest moduli the limb form takes (tiny1024's)
N = 1024


@pytest.fixture(scope="module")
def eng():
    import crcnn_amd as ca
    E = ca.Engine(N, Q, 1 << 20, device=0)
    yield E, ca
    E.close()


def rand_rows(rng, E, rows, edge=False):
    """[rows][k][n] uniform residues; edge: only the values whose centred digits are extreme (0, 1, q-1, q/2, q/2+1)"""
    out = np.empty((rows, E.k, E.n), dtype=np.uint64)
    for i, q in enumerate(E.q):
        if edge:
            out[:, i] = rng.choice(np.array([0, 1, q - 1, q // 2, q // 2 + 1, 0x7f7f7f7f7f7f7f % q, 0x80808080808080 % q], dtype=np.uint64), size=(rows, E.n))
        else:
            out[:, i] = rng.integers(0, q, size=(rows, E.n), dtype=np.uint64)
    return out


def vector_alu_conv(E, ca, d_x, d_w, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf):
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    d_y = E.alloc(B * nf * xo * yo * 2 * E.k * E.n * 8)
    d_work = E.alloc(E.conv2d_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT))
    E.conv2d(d_x, d_w, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTT, d_y, d_work)
    E.sync()
    return E.download(d_y, (B * nf * xo * yo * 2 * E.k, E.n))


# ---- the oracle as the direct checker -----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def O():
    from oracle import orc
    return orc.Oracle(N, Q, 1 << 20)


def oracle_layer(O, x_ntt, w_ntt, bias_vals, B, zd, xd, yd, xs, ys, xf, yf, nf):
    """the reference's convolution (oracle restatement) on operands whose NTT-DOMAIN values are the given rows: the input ciphertexts are the oracle's own inverse
    transforms of x_ntt, so that the products inside are exactly x_ntt * w_ntt.  Returns (coefficient-form inputs, coefficient-form result, NTT-form result)."""
    k, n = O.k, O.n
    xc = np.empty((B * zd * xd * yd, 2, k, n), dtype=np.uint64)
    for i, ct in enumerate(x_ntt.reshape(-1, 2, k, n)):
        xc[i] = O.ct_from_ntt(ct)
    xc = xc.reshape(B, zd, xd, yd, 2, k, n)
    w = np.ascontiguousarray(w_ntt.reshape(nf, zd, xf, yf, k, n))
    bp = O.encode_many(bias_vals)
    macs = nf * ((xd - xf) // xs + 1) * ((yd - yf) // ys + 1) * zd * xf * yf
    # reference operation order (one inverse transform per product) where that takes seconds, NTT-domain accumulation of the same oracle primitives otherwise
    # (bit-identical: tests/test_oracle_layers.py::test_conv)
    want = np.stack([O.conv(xc[b], w, bp, xs, ys, threads=8, fast=macs > 20000) for b in range(B)])
    want_ntt = np.empty_like(want)
    flat, flat_n = want.reshape(-1, 2, k, n), want_ntt.reshape(-1, 2, k, n)
    for i in range(flat.shape[0]):
        flat_n[i] = O.ct_to_ntt(flat[i])
    return xc, want, want_ntt


def np_limb_tensor(y_ntt, q, dense):
    """numpy statement of the limb operand form (include/crcnn_hip.h CRC_NTTL): y_ntt [B][C][P][2][k][n] canonical residues -> int8 balanced base-256 digits of the
    centred representatives; conv consumer [k][n][B][7][P][2][C^32], dense consumer (all C * P outputs as channels of one position) [k][n][7][ch/32][B*2][32]"""
    B, C, P = y_ntt.shape[:3]
    k, n = y_ntt.shape[-2:]
    qv = np.array(q, dtype=np.int64).reshape(1, 1, 1, 1, k, 1)
    v = y_ntt.astype(np.int64)
    v = np.where(v > (qv >> 1), v - qv, v)
    digs = []
    for _ in range(7):
        d = ((v + 128) & 255) - 128
        digs.append(d.astype(np.int8)); v = (v - d) >> 8
    assert not v.any()
    dg = np.stack(digs)                                   # [7][B][C][P][2][k][n]
    if not dense and C < 32:
        # flat form (round 4: fewer than 32 channels): [k][n][B][7][2][P][zdc], zdc = C rounded up to 4, every image padded to a multiple of 16 bytes
        zdc = -(-C // 4) * 4; img = -(-(7 * 2 * P * zdc) // 16) * 16
        out = np.zeros((k, n, B, img), dtype=np.int8)
        out[..., :7 * 2 * P * zdc].reshape(k, n, B, 7, 2, P, zdc)[..., :C] = dg.transpose(5, 6, 1, 0, 4, 3, 2)
        return out
    if not dense:
        Cp = -(-C // 32) * 32
        out = np.zeros((k, n, B, 7, P, 2, Cp), dtype=np.int8)
        out[..., :C] = dg.transpose(5, 6, 1, 0, 3, 4, 2)
        return out
    ch = C * P; chp = -(-ch // 32) * 32
    flat = np.zeros((7, B, chp, 2, k, n), dtype=np.int8)
    flat[:, :, :ch] = dg.reshape(7, B, ch, 2, k, n)
    # [k][n][7][chp/32][B][2][32]
    return np.ascontiguousarray(flat.reshape(7, B, chp // 32, 32, 2, k, n).transpose(5, 6, 0, 2, 1, 4, 3))


def limb_defined(t, C, P, dense=False):
    """the bytes of a limb tensor that its producers write (a flat convolution tensor pads every image to 16 bytes: the padding is never read with a non-zero weight)"""
    if dense or C >= 32 or t.ndim != 4:
        return t
    zdc = -(-C // 4) * 4
    return t[..., :7 * 2 * P * zdc]


def upload_limb(E, xl, nbytes):
    """a limb tensor made in numpy into a device buffer of the size the engine asks for (the last 32-byte piece of a flat run may be read past the last position)"""
    d = E.alloc(nbytes)
    E.L.crc_memset(E.c, E.p(d), 0, nbytes, E.stream)
    h = E.upload(xl)
    E.L.crc_memcpy_d2d(E.c, E.p(d), E.p(h), xl.nbytes, E.stream); E.sync()
    return d


def gpu_bias(E, ca, bias_vals, form):
    pl, _ = E.encode(np.asarray(bias_vals, dtype=np.float32))
    d_b = E.alloc(len(pl) * E.k * E.n * 8)
    E.plain_to_delta(E.upload(pl), len(pl), form, d_b)
    return d_b


ORACLE_CONV1_SHAPES = [
    # xd, yd, xs, ys, xf, yf, nf, B        (zd = 1)
    (28, 28, 2, 2, 6, 6, 32, 1),           # PlainModelTiny conv1 + pool1
    (28, 28, 2, 2, 7, 7, 20, 1),           # ApproxPlainModel / PlainModelWoPad conv1 + pool1
    (9, 11, 1, 2, 3, 4, 5, 2),             # ragged
    (8, 8, 1, 1, 8, 8, 1, 3),              # a single output pixel and filter
]


@pytest.mark.parametrize("shape", ORACLE_CONV1_SHAPES)
@pytest.mark.parametrize("edge", [False, True])
def test_conv1_kernel_equals_oracle(eng, O, shape, edge):
    E, ca = eng
    xd, yd, xs, ys, xf, yf, nf, B = shape
    rng = np.random.default_rng(xd * 999 + yf * 13 + nf + 7 * edge)
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    x = rand_rows(rng, E, B * xd * yd * 2, edge); w = rand_rows(rng, E, nf * xf * yf, edge)
    bv = rng.normal(0, 0.3, nf).astype(np.float32)
    xc, want, want_ntt = oracle_layer(O, x, w, bv, B, 1, xd, yd, xs, ys, xf, yf, nf)
    d_w = E.upload(w)
    d_wl = E.alloc(E.limb_conv1_weights_bytes()); E.limb_conv1_pack_weights(d_w, nf, xf, yf, d_wl)
    rows_y = B * nf * xo * yo * 2 * E.k
    d_y = E.alloc(max(rows_y * E.n * 8, E.limb_tensor_bytes(B, nf, xo, yo)))
    # the layer as the reference calls it: coefficient form in and out
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL1, ca.COEFF))
    E.conv2d(E.upload(xc), d_wl, gpu_bias(E, ca, bv, ca.COEFF), B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.COEFF, d_y, d_work, w_form=ca.NTTL1)
    E.sync()
    assert np.array_equal(E.download(d_y, want.shape), want), (shape, edge, "coefficient form")
    # NTT-resident
    d_bn = gpu_bias(E, ca, bv, ca.NTT)
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL1, ca.NTT))
    E.conv2d(E.upload(x), d_wl, d_bn, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTT, d_y, d_work, w_form=ca.NTTL1)
    E.sync()
    assert np.array_equal(E.download(d_y, want.shape), want_ntt), (shape, edge, "NTT form")
    # hand-over to a matrix-core convolution: the limb tensor, built here from the oracle's result
    if xo * yo > 1:
        nb = E.limb_tensor_bytes(B, nf, xo, yo)
        E.L.crc_memset(E.c, E.p(d_y), 0, nb, E.stream)
        d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL1, ca.NTTLC))
        E.conv2d(E.upload(x), d_wl, d_bn, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTLC, d_y, d_work, w_form=ca.NTTL1)
        E.sync()
        exp = np_limb_tensor(want_ntt.reshape(B, nf, xo * yo, 2, E.k, E.n), E.q, dense=False)
        assert np.array_equal(limb_defined(E.download(d_y, exp.shape, dtype=np.int8), nf, xo * yo), limb_defined(exp, nf, xo * yo)), (shape, edge, "limb tensor")


ORACLE_GEMM_SHAPES = [
    # zd, xd, yd, xs, ys, xf, yf, nf, B
    (32, 6, 6, 1, 1, 3, 3, 64, 2),          # 4 x 4 pixels: direct dense hand-over (2P | 64)
    (20, 5, 7, 2, 1, 3, 2, 50, 1),          # channel and filter padding, ragged rows; 2P = 24: slot-major result + conversion kernel.  20 channels: flat form, 2 steps per window row
    (20, 11, 11, 2, 2, 3, 3, 50, 2),        # ApproxPlainModel / PlainModelWoPad conv2 (flat form: 3 x 20 = 60 bytes per window row, 6 reduction steps instead of 10)
    (24, 6, 6, 1, 1, 3, 3, 40, 2),          # 24 channels: 72 bytes per window row, 3 steps of which the last holds 8 terms; 4 x 4 pixels: direct dense hand-over
    (6, 7, 6, 1, 2, 2, 3, 33, 3),           # 6 channels in 8 bytes per position: one 32-byte step holds a whole 3-tap window row; odd step count (zero step)
    (70, 1, 1, 1, 1, 1, 1, 10, 9),          # a dense layer (dense -> dense hand-over)
]


@pytest.mark.parametrize("variant", [2, 1], ids=["two-workgroups-per-CU", "one-workgroup-per-CU"])
@pytest.mark.parametrize("shape", ORACLE_GEMM_SHAPES)
@pytest.mark.parametrize("edge", [False, True])
def test_limb_gemm_equals_oracle(eng, O, shape, edge, variant, request):
    E, ca = eng
    E.set_tuning("mfma_variant", variant); request.addfinalizer(lambda: E.set_tuning("mfma_variant", 2))
    zd, xd, yd, xs, ys, xf, yf, nf, B = shape
    rng = np.random.default_rng(zd * 101 + nf + 5 * edge)
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    x = rand_rows(rng, E, B * zd * xd * yd * 2, edge); w = rand_rows(rng, E, nf * zd * xf * yf, edge)
    bv = rng.normal(0, 0.3, nf).astype(np.float32)
    xc, want, want_ntt = oracle_layer(O, x, w, bv, B, zd, xd, yd, xs, ys, xf, yf, nf)
    d_wl = E.alloc(E.limb_weights_bytes(nf, zd, xf, yf)); E.limb_pack_weights(E.upload(w), nf, zd, xf, yf, d_wl)
    d_y = E.alloc(max(want.nbytes, E.limb_tensor_bytes(B, nf * xo * yo)))
    # coefficient form in and out
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL, ca.COEFF))
    E.conv2d(E.upload(xc), d_wl, gpu_bias(E, ca, bv, ca.COEFF), B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.COEFF, d_y, d_work, w_form=ca.NTTL)
    E.sync()
    assert np.array_equal(E.download(d_y, want.shape), want), (shape, edge, "coefficient form")
    # limb tensor in (made in numpy from the same rows: pins crc_limb_pack_tensor's layout too), NTT form out
    d_bn = gpu_bias(E, ca, bv, ca.NTT)
    dense_in = xd * yd == 1
    xl = np_limb_tensor(x.reshape(B, zd, xd * yd, 2, E.k, E.n), E.q, dense=dense_in)
    d_xl = E.alloc(E.limb_tensor_bytes(B, zd, xd, yd)); E.limb_pack_tensor(E.upload(x), ca.NTT, B, zd, xd, yd, d_xl); E.sync()
    assert np.array_equal(limb_defined(E.download(d_xl, xl.shape, dtype=np.int8), zd, xd * yd, dense_in), limb_defined(xl, zd, xd * yd, dense_in)), (shape, edge, "crc_limb_pack_tensor")
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTTL, ca.NTTL, ca.NTT))
    E.conv2d(upload_limb(E, xl, E.limb_tensor_bytes(B, zd, xd, yd)), d_wl, d_bn, B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTTL, ca.NTT, d_y, d_work, w_form=ca.NTTL)
    E.sync()
    assert np.array_equal(E.download(d_y, want.shape), want_ntt), (shape, edge, "limb in, NTT out")
    # hand-over to a dense matrix-core layer
    nb = E.limb_tensor_bytes(B, nf * xo * yo)
    E.L.crc_memset(E.c, E.p(d_y), 0x55, nb, E.stream)
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL, ca.NTTL))
    E.conv2d(E.upload(x), d_wl, d_bn, B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL, d_y, d_work, w_form=ca.NTTL)
    E.sync()
    exp = np_limb_tensor(want_ntt.reshape(B, nf, xo * yo, 2, E.k, E.n), E.q, dense=True)
    assert np.array_equal(E.download(d_y, exp.shape, dtype=np.int8), exp), (shape, edge, "dense hand-over")


def test_reference_layer_outputs_with_limb_weights():
    """CrCNN's own ConvolutionalLayer / FullyConnectedLayer outputs (tests/golden/layers_n256_k2_t20.npz, layers1_...: compiled reference) reproduced with the
    weights in the matrix-core forms: CRC_NTTL for the two-channel convolution and the dense layer, CRC_NTTL1 for the one-channel convolution"""
    import os
    import crcnn_amd as ca
    gd = os.path.join(os.path.dirname(__file__), "golden")
    g = dict(np.load(os.path.join(gd, "layers_n256_k2_t20.npz")))
    E = ca.Engine(int(g["n"]), [int(v) for v in g["q"]], int(g["t"]), device=0)

    def weights(wv):
        pl, _ = E.encode(np.asarray(wv, dtype=np.float32)); d_w = E.alloc(len(pl) * E.k * E.n * 8)
        E.plain_to_ntt(E.upload(pl), len(pl), d_w); return d_w

    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    for variant in (2, 1):
        E.set_tuning("mfma_variant", variant)
        assert E.limb_supported(zd, xf, yf)
        d_wl = E.alloc(E.limb_weights_bytes(nf, zd, xf, yf)); E.limb_pack_weights(weights(g["conv_w"]), nf, zd, xf, yf, d_wl)
        for B in (1, 3):
            x = np.ascontiguousarray(np.repeat(g["x"][None], B, axis=0))
            d_y = E.alloc(B * nf * xo * yo * 2 * E.k * E.n * 8)
            d_work = E.alloc(E.conv2d_forms_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL, ca.COEFF))
            E.conv2d(E.upload(x), d_wl, gpu_bias(E, ca, g["conv_b"], ca.COEFF), B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.COEFF, d_y, d_work, w_form=ca.NTTL)
            y = E.download(d_y, (B, nf, xo, yo, 2, E.k, E.n))
            for b in range(B):
                assert np.array_equal(y[b], g["ref_conv"]), ("conv", variant, B)
        out_dim, in_dim = g["fc_w"].shape
        assert E.limb_supported(in_dim, 1, 1)
        d_wl = E.alloc(E.limb_weights_bytes(out_dim, in_dim, 1, 1)); E.limb_pack_weights(weights(g["fc_w"]), out_dim, in_dim, 1, 1, d_wl)
        d_y = E.alloc(out_dim * 2 * E.k * E.n * 8)
        d_work = E.alloc(E.conv2d_forms_work_bytes(1, in_dim, 1, 1, 1, 1, 1, 1, out_dim, ca.COEFF, ca.NTTL, ca.COEFF))
        E.dense(E.upload(g["x"]), d_wl, gpu_bias(E, ca, g["fc_b"], ca.COEFF), 1, in_dim, out_dim, ca.COEFF, ca.COEFF, d_y, d_work, w_form=ca.NTTL)
        assert np.array_equal(E.download(d_y, (1, out_dim, 1, 2, E.k, E.n)), g["ref_fc"]), ("fc", variant)
    E.set_tuning("mfma_variant", 2)
    g1 = dict(np.load(os.path.join(gd, "layers1_n256_k2_t20.npz")))
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g1["dims"][:8]]
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    assert zd == 1 and E.limb_conv1_supported(1, xd, yd, xs, ys, xf, yf, nf)
    d_wl = E.alloc(E.limb_conv1_weights_bytes()); E.limb_conv1_pack_weights(weights(g1["conv_w"]), nf, xf, yf, d_wl)
    for B in (1, 2):
        x = np.ascontiguousarray(np.repeat(g1["x"][None], B, axis=0))
        d_y = E.alloc(B * nf * xo * yo * 2 * E.k * E.n * 8)
        d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL1, ca.COEFF))
        E.conv2d(E.upload(x), d_wl, gpu_bias(E, ca, g1["conv_b"], ca.COEFF), B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.COEFF, d_y, d_work, w_form=ca.NTTL1)
        y = E.download(d_y, (B, nf, xo, yo, 2, E.k, E.n))
        for b in range(B):
            assert np.array_equal(y[b], g1["ref_conv"]), ("one-channel conv", B)
    E.close()


CONV1_SHAPES = [
    # xd, yd, xs, ys, xf, yf, nf, B        (zd = 1)
    (28, 28, 2, 2, 6, 6, 32, 3),           # PlainModelTiny conv1 + pool1
    (28, 28, 2, 2, 7, 7, 20, 2),           # ApproxPlainModel / PlainModelWoPad conv1 + pool1
    (9, 11, 1, 2, 3, 4, 5, 1),             # ragged: 7 x 4 outputs (one partial row tile), window shorter than 8 both ways
    (12, 10, 3, 1, 8, 8, 32, 2),           # full 8 x 8 window
    (8, 8, 1, 1, 8, 8, 1, 5),              # a single output pixel and filter
    (10, 9, 1, 1, 3, 3, 17, 2),            # 17 filters (one leftover), 8 x 7 outputs = 7 row tiles: the leftover waves' last group of four is ragged
    (13, 13, 1, 1, 3, 3, 19, 1),           # 19 filters, 11 x 11 outputs = 16 row tiles (four full groups); one image
    (7, 6, 1, 1, 4, 4, 20, 3),             # 20 filters, 4 x 3 outputs = 2 row tiles: fewer tiles than one group
]


@pytest.mark.parametrize("shape", CONV1_SHAPES)
@pytest.mark.parametrize("edge", [False, True])
def test_conv1_matrix_core_kernel(eng, shape, edge):
    E, ca = eng
    xd, yd, xs, ys, xf, yf, nf, B = shape
    assert E.limb_conv1_supported(1, xd, yd, xs, ys, xf, yf, nf)
    rng = np.random.default_rng(xd * 1000 + yf * 10 + nf + edge)
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    x = rand_rows(rng, E, B * xd * yd * 2, edge); w = rand_rows(rng, E, nf * xf * yf, edge); b = rand_rows(rng, E, nf)
    d_x, d_w, d_b = E.upload(x), E.upload(w), E.upload(b)
    want = vector_alu_conv(E, ca, d_x, d_w, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf)
    d_wl = E.alloc(E.limb_conv1_weights_bytes()); E.limb_conv1_pack_weights(d_w, nf, xf, yf, d_wl)
    rows_y = B * nf * xo * yo * 2 * E.k
    d_y = E.alloc(max(rows_y * E.n * 8, E.limb_tensor_bytes(B, nf, xo, yo)))
    d_xp = E.upload(x); E.pack28(d_xp, B * xd * yd * 2 * E.k)
    # 17-20 filters: the second filter group runs packed (round 5: (filter, weight limb) in the rows of the MFMA's A operand, CRC_CONV1_NARROW) or like a full
    # one (0: the round-4 form) -- the same results either way
    forms = [(ca.NTT, ca.NTT, 1), (ca.NTTP, ca.NTTP, 1), (ca.NTT, ca.NTTLC, 1)] + ([(ca.NTT, ca.NTT, 0), (ca.NTT, ca.NTTLC, 0)] if 16 < nf <= 20 else [])
    for fin, fout, nar in forms:
        E.set_tuning("conv1_narrow", nar)
        d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, fin, ca.NTTL1, fout))
        if fout == ca.NTTLC:
            E.L.crc_memset(E.c, E.p(d_y), 0, E.limb_tensor_bytes(B, nf, xo, yo), E.stream)       # (the tensor's tail -- read-ahead room of the flat form -- is nobody's to write)
        E.conv2d(d_xp if fin == ca.NTTP else d_x, d_wl, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, fin, fout, d_y, d_work, w_form=ca.NTTL1)
        if fout == ca.NTTP:
            E.pack28(d_y, rows_y, unpack=True)
        if fout == ca.NTTLC:
            # the limb tensor a convolution reads: must be what crc_limb_pack_tensor makes of the NTT-form result
            d_ref = E.alloc(E.limb_tensor_bytes(B, nf, xo, yo)); E.L.crc_memset(E.c, E.p(d_ref), 0, E.limb_tensor_bytes(B, nf, xo, yo), E.stream)
            E.limb_pack_tensor(E.upload(want.reshape(-1)), ca.NTT, B, nf, xo, yo, d_ref)
            E.sync()
            nb = E.limb_tensor_bytes(B, nf, xo, yo)
            assert np.array_equal(E.download(d_y, (nb // 8,)), E.download(d_ref, (nb // 8,))), (shape, "limb tensor")
        else:
            E.sync()
            assert np.array_equal(E.download(d_y, (rows_y, E.n)), want), (shape, fin, fout, nar)
    E.set_tuning("conv1_narrow", 1)
    # coefficient-form input and output (the layer as the reference calls it): INTT of the NTT-form result
    d_xc = E.upload(x); E.ntt_inv(d_xc, B * xd * yd)
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL1, ca.NTT))
    E.conv2d(d_xc, d_wl, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTT, d_y, d_work, w_form=ca.NTTL1)
    E.sync()
    assert np.array_equal(E.download(d_y, (rows_y, E.n)), want), (shape, "coeff in")


def test_conv1_multi_pass(eng, request):
    """a batch whose work space would exceed the per-pass cap is processed in sub-batches: same ciphertexts, same limb tensor"""
    E, ca = eng
    xd, yd, xs, ys, xf, yf, nf, B = 28, 28, 2, 2, 6, 6, 32, 5
    xo, yo = 12, 12
    rng = np.random.default_rng(5)
    x = rand_rows(rng, E, B * xd * yd * 2); w = rand_rows(rng, E, nf * xf * yf); b = rand_rows(rng, E, nf)
    d_x, d_w, d_b = E.upload(x), E.upload(w), E.upload(b)
    want = vector_alu_conv(E, ca, d_x, d_w, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf)
    d_wl = E.alloc(E.limb_conv1_weights_bytes()); E.limb_conv1_pack_weights(d_w, nf, xf, yf, d_wl)
    rows_y = B * nf * xo * yo * 2 * E.k
    nb = E.limb_tensor_bytes(B, nf, xo, yo)
    d_y = E.alloc(max(rows_y * E.n * 8, nb))
    whole = E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL1, ca.NTT)
    E.set_tuning("conv1_pass_bytes", whole // 3)          # a third of the whole: one or two images per pass
    request.addfinalizer(lambda: E.set_tuning("conv1_pass_bytes", 0))
    assert E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL1, ca.NTT) < whole // 2
    d_xc = E.upload(x); E.ntt_inv(d_xc, B * xd * yd)
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTTL1, ca.NTT))
    E.conv2d(d_xc, d_wl, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.NTT, d_y, d_work, w_form=ca.NTTL1)
    E.sync()
    assert np.array_equal(E.download(d_y, (rows_y, E.n)), want)
    d_work2 = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL1, ca.NTTLC))        # (a pass holds more images when no u64 result is staged)
    E.L.crc_memset(E.c, E.p(d_y), 0, nb, E.stream)             # (the tensor's last 64 bytes are read-ahead room that no producer writes)
    E.conv2d(d_x, d_wl, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTLC, d_y, d_work2, w_form=ca.NTTL1)
    d_ref = E.alloc(nb); E.L.crc_memset(E.c, E.p(d_ref), 0, nb, E.stream)
    E.limb_pack_tensor(E.upload(want.reshape(-1)), ca.NTT, B, nf, xo, yo, d_ref)
    E.sync()
    assert np.array_equal(E.download(d_y, (nb // 8,)), E.download(d_ref, (nb // 8,)))


def test_conv1_shapes_outside_the_kernel_are_refused(eng):
    E, ca = eng
    assert not E.limb_conv1_supported(2, 28, 28, 2, 2, 6, 6, 32)        # more than one channel
    assert not E.limb_conv1_supported(1, 28, 28, 1, 1, 9, 5, 32)        # window taller than 8
    assert not E.limb_conv1_supported(1, 28, 28, 2, 2, 6, 6, 33)        # more than 32 filters
    assert not E.limb_conv1_supported(1, 28, 28, 1, 1, 5, 5, 32)        # 24 x 24 outputs: image + staging exceed the LDS
    d = E.alloc(1 << 20)
    with pytest.raises(ca.CrcError):
        E.conv2d(d, d, d, 1, 2, 28, 28, 2, 2, 6, 6, 32, ca.NTT, ca.NTT, d, d, w_form=ca.NTTL1)


GEMM_SHAPES = [
    # zd, xd, yd, xs, ys, xf, yf, nf, B
    (32, 6, 6, 1, 1, 3, 3, 64, 2),
    (20, 5, 7, 2, 1, 3, 2, 50, 1),         # channel and filter padding, ragged rows
    (70, 1, 1, 1, 1, 1, 1, 10, 9),         # a dense layer
]


@pytest.mark.parametrize("variant", ["2", "1"], ids=["two-workgroups-per-CU", "one-workgroup-per-CU"])
@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_limb_gemm_kernel(eng, shape, variant, request):
    E, ca = eng
    E.set_tuning("mfma_variant", int(variant)); request.addfinalizer(lambda: E.set_tuning("mfma_variant", 2))
    zd, xd, yd, xs, ys, xf, yf, nf, B = shape
    rng = np.random.default_rng(zd * 100 + nf)
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    x = rand_rows(rng, E, B * zd * xd * yd * 2); w = rand_rows(rng, E, nf * zd * xf * yf, edge=True); b = rand_rows(rng, E, nf)
    d_x, d_w, d_b = E.upload(x), E.upload(w), E.upload(b)
    want = vector_alu_conv(E, ca, d_x, d_w, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf)
    d_wl = E.alloc(E.limb_weights_bytes(nf, zd, xf, yf)); E.limb_pack_weights(d_w, nf, zd, xf, yf, d_wl)
    rows_y = B * nf * xo * yo * 2 * E.k
    d_y = E.alloc(rows_y * E.n * 8)
    d_xl = E.alloc(E.limb_tensor_bytes(B, zd, xd, yd)); E.limb_pack_tensor(d_x, ca.NTT, B, zd, xd, yd, d_xl)
    for fin, fout in [(ca.NTT, ca.NTT), (ca.NTTL, ca.NTTP)]:
        d_work = E.alloc(E.conv2d_forms_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, fin, ca.NTTL, fout))
        E.conv2d(d_xl if fin == ca.NTTL else d_x, d_wl, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf, fin, fout, d_y, d_work, w_form=ca.NTTL)
        if fout == ca.NTTP:
            E.pack28(d_y, rows_y, unpack=True)
        E.sync()
        assert np.array_equal(E.download(d_y, (rows_y, E.n)), want), (shape, fin, fout)


DENSE_HANDOVER_SHAPES = [
    # zd, xd, yd, xs, ys, xf, yf, nf, B      the layer's result goes to a dense layer in limb form (out_form = CRC_NTTL), written by the GEMM kernel itself when 2P | 64
    (32, 6, 6, 1, 1, 3, 3, 64, 3),          # 4 x 4 pixels (PlainModelTiny's conv2+pool2 shape): a tile = two images, the last tile ragged
    (32, 5, 5, 1, 1, 2, 2, 50, 2),          # 50 filters: the tile's filters past 50 are zero padding of the consumer's channels
    (70, 1, 1, 1, 1, 1, 1, 10, 9),          # dense -> dense: 32 images per tile, 10 of 32 padded channels
    (32, 5, 5, 1, 1, 3, 3, 64, 2),          # 3 x 3 pixels: 2P = 18 does not divide 64 -> slot-major result + conversion kernel
]


@pytest.mark.parametrize("variant", ["2", "1"], ids=["two-workgroups-per-CU", "one-workgroup-per-CU"])
@pytest.mark.parametrize("shape", DENSE_HANDOVER_SHAPES)
def test_limb_gemm_hands_over_to_dense(eng, shape, variant, request):
    E, ca = eng
    E.set_tuning("mfma_variant", int(variant)); request.addfinalizer(lambda: E.set_tuning("mfma_variant", 2))
    zd, xd, yd, xs, ys, xf, yf, nf, B = shape
    rng = np.random.default_rng(zd * 7 + nf)
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    x = rand_rows(rng, E, B * zd * xd * yd * 2); w = rand_rows(rng, E, nf * zd * xf * yf); b = rand_rows(rng, E, nf)
    d_x, d_w, d_b = E.upload(x), E.upload(w), E.upload(b)
    want = vector_alu_conv(E, ca, d_x, d_w, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf)
    d_wl = E.alloc(E.limb_weights_bytes(nf, zd, xf, yf)); E.limb_pack_weights(d_w, nf, zd, xf, yf, d_wl)
    nb = E.limb_tensor_bytes(B, nf * xo * yo)          # the consumer's tensor: nf * P channels, one position
    d_y = E.alloc(nb); E.L.crc_memset(E.c, E.p(d_y), 0x55, nb, E.stream)
    d_work = E.alloc(E.conv2d_forms_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL, ca.NTTL))
    E.conv2d(d_x, d_wl, d_b, B, zd, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL, d_y, d_work, w_form=ca.NTTL)
    d_ref = E.alloc(nb); E.L.crc_memset(E.c, E.p(d_ref), 0, nb, E.stream)
    E.limb_pack_tensor(E.upload(want.reshape(-1)), ca.NTT, B, nf * xo * yo, 1, 1, d_ref)
    E.sync()
    assert np.array_equal(E.download(d_y, (nb // 8,)), E.download(d_ref, (nb // 8,))), shape
