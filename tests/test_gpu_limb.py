"""GPU parity, layer level, of the matrix-core kernels (kernels_mfma.hip / kernels_mfma1.hip) against the vector-ALU kernel of the same layer, which
tests/test_gpu_layers.py pins to the reference's own Layer::forward outputs.  Random full-range residues (the worst case for the limb split: digits at
both ends of [-128, 127]), every operand / result form, ragged shapes.  Bit-exact."""
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


CONV1_SHAPES = [
    # xd, yd, xs, ys, xf, yf, nf, B        (zd = 1)
    (28, 28, 2, 2, 6, 6, 32, 3),           # PlainModelTiny conv1 + pool1
    (28, 28, 2, 2, 7, 7, 20, 2),           # ApproxPlainModel / PlainModelWoPad conv1 + pool1
    (9, 11, 1, 2, 3, 4, 5, 1),             # ragged: 7 x 4 outputs (one partial row tile), window shorter than 8 both ways
    (12, 10, 3, 1, 8, 8, 32, 2),           # full 8 x 8 window
    (8, 8, 1, 1, 8, 8, 1, 5),              # a single output pixel and filter
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
    for fin, fout in [(ca.NTT, ca.NTT), (ca.NTTP, ca.NTTP), (ca.NTT, ca.NTTLC)]:
        d_work = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, fin, ca.NTTL1, fout))
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
            assert np.array_equal(E.download(d_y, (rows_y, E.n)), want), (shape, fin, fout)
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
