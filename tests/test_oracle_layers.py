"""Oracle layer loops against the reference's own Layer::forward (CrCNN/src/*Layer.cpp compiled in oracle/_ref)."""
import os

import numpy as np
import pytest

from oracle import orc

G = os.path.join(os.path.dirname(__file__), "golden", "layers_n256_k2_t20.npz")


@pytest.fixture(scope="module")
def gl():
    g = dict(np.load(G))
    O = orc.Oracle(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]))
    return g, O


def test_layer_inputs_reproducible(gl):
    g, O = gl
    sk, pk = O.keygen(3000)
    assert np.array_equal(sk, g["sk"]) and np.array_equal(O.gen_evk(3001, sk), g["evk"])
    zd, xd, yd = g["img"].shape
    x = O.encrypt_many(pk, O.encode_many(g["img"]).reshape(zd, xd, yd, O.n), 4000)
    assert np.array_equal(x, g["x"])


def enc_w(O, w):
    return O.plains_to_ntt(O.encode_many(w).reshape(w.shape + (O.n,)))


def test_conv(gl):
    g, O = gl
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    w = enc_w(O, g["conv_w"]); b = O.encode_many(g["conv_b"])
    for threads in (1, 3):
        y = O.conv(g["x"], w, b, xs, ys, threads=threads)
        assert np.array_equal(y, g["ref_conv"])
    # NTT-domain accumulation (one INTT per output) is bit-identical to the reference order
    assert np.array_equal(O.conv(g["x"], w, b, xs, ys, fast=True), g["ref_conv"])
    # decrypted result equals the float convolution
    for f in range(nf):
        want = (g["img"][:, 0:xf, 0:yf].astype(np.float64) * g["conv_w"][f]).sum() + g["conv_b"][f]
        assert abs(O.decrypt_value(g["sk"], y[f, 0, 0]) - want) < 1e-5


def test_one_channel_conv_matches_reference_layer():
    """the reference's ConvolutionalLayer on a ONE-channel input (conv1's shape class; tests/golden/layers1_n256_k2_t20.npz): oracle, both operation orders"""
    g = dict(np.load(os.path.join(os.path.dirname(G), "layers1_n256_k2_t20.npz")))
    O = orc.Oracle(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]))
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    assert zd == 1
    w = enc_w(O, g["conv_w"]); b = O.encode_many(g["conv_b"])
    assert np.array_equal(O.conv(g["x"], w, b, xs, ys, threads=2), g["ref_conv"])
    assert np.array_equal(O.conv(g["x"], w, b, xs, ys, fast=True), g["ref_conv"])


def test_fc(gl):
    g, O = gl
    w = enc_w(O, g["fc_w"]); b = O.encode_many(g["fc_b"])
    y = O.fc(g["x"], w, b, threads=2)
    assert np.array_equal(y, g["ref_fc"])
    want = g["fc_w"].astype(np.float64) @ g["img"].reshape(-1).astype(np.float64) + g["fc_b"]
    got = [O.decrypt_value(g["sk"], y[0, i, 0]) for i in range(len(want))]
    assert np.allclose(got, want, atol=1e-5)


def test_pools(gl):
    g, O = gl
    pxs, pys, pxf, pyf = [int(v) for v in g["dims"][9:13]]
    assert np.array_equal(O.pool(g["x"], pxs, pys, pxf, pyf), g["ref_pool"])
    div, _ = O.encode(1.0 / (pxf * pyf))                      # avgPoolingLayer.cpp:12
    assert np.array_equal(O.pool(g["x"], pxs, pys, pxf, pyf, div_plain=div, threads=2), g["ref_avgpool"])


def test_batchnorm(gl):
    g, O = gl
    mean = O.encode_many(g["bn_mean"])
    invstd = np.float32(1.0 / np.sqrt(g["bn_var"].astype(np.float64) + 0.00001))    # cnnBuilder.cpp:100-102
    y = O.bn(g["x"], mean, O.encode_many(invstd), threads=2)
    assert np.array_equal(y, g["ref_bn"])


def test_square_layer(gl):
    g, O = gl
    y = O.square_layer(g["x"], g["evk"], threads=3)
    assert np.array_equal(y, g["ref_square"])
    v = float(g["img"][1, 2, 3])
    assert abs(O.decrypt_value(g["sk"], y[1, 2, 3]) - v * v) < 1e-4
