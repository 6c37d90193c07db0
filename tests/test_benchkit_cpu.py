"""bench.py's helpers that need no GPU (tools/benchkit): the fused layer plan must be the layer list the C++ classes' Network::fuse() produces (bench.py matches
bench_host's names against it before it prices the dominant layer), and the executed / useful ratio of the limb GEMM must follow the kernel's tiling."""
import numpy as np

import crcnn_amd as ca
from benchkit import geometry
from benchkit.configs import CONFIGS


def test_fused_plan_names_and_shapes():
    want = {
        "tiny4096": ["pool1_features.conv1+pool1", "pool2_features.conv2+pool2", "classifier.fc3", "classifier.fc4"],
        "approx8192": ["pool1_features.conv1+pool1", "pool1_features.norm1+pool2_features.conv2", "act1+pool2", "pool2_features.norm2+classifier.fc3", "classifier.fc4"],
        "wopad16384": ["pool1_features.conv1+pool1", "pool1_features.norm1+pool2_features.conv2", "act1+pool2", "pool2_features.norm2+classifier.fc3", "classifier.fc4"],
    }
    for name, layers in want.items():
        cfg = CONFIGS[name]
        E = ca.Engine(cfg["n"], ca.default_coeff_modulus_128(cfg["n"])[:cfg["k"]], cfg["t"], device=-1)
        plan = geometry.fused_plan(E, cfg["model"])
        assert [pl[1] for pl in plan] == layers, name
        assert plan[0][3] == (1, 28, 28) and plan[-1][4] == (1, 10, 1)
        for a, b in zip(plan, plan[1:]):
            assert int(np.prod(a[4])) == int(np.prod(b[3])), (name, a[1], b[1])         # every layer reads what the one in front wrote
        E.close()
    # the host decides from the HBM it finds and from what the key switch holds: a layer list without the Square + pooling pair is followed as it is
    cfg = CONFIGS["approx8192"]
    E = ca.Engine(cfg["n"], ca.default_coeff_modulus_128(cfg["n"])[:cfg["k"]], cfg["t"], device=-1)
    unpaired = ["pool1_features.conv1+pool1", "pool1_features.norm1+pool2_features.conv2", "act1", "pool2", "pool2_features.norm2+classifier.fc3", "classifier.fc4"]
    assert [pl[1] for pl in geometry.fused_plan(E, cfg["model"], names=unpaired)] == unpaired
    E.close()


def test_limb_exec_over_useful_follows_the_tiling():
    # ApproxPlainModel's conv2: 20 channels, 3 x 3, 50 filters, 5 x 5 outputs; 32 images per launch = 1600 rows (25 tiles of 64), flat form: 6 steps of 32 for 180 terms
    a = dict(zd=20, xd=11, yd=11, xs=2, ys=2, xf=3, yf=3, nf=50)
    r = geometry.limb_exec_over_useful("conv", a, 32, 50 * 25)
    assert abs(r - (1.0 * (6 * 32 / 180) * (64 / 50))) < 1e-12
    # PlainModelTiny's conv2+pool2: 32 channels, 6 x 6, 64 filters, 16 outputs, 128 images: no padding at all
    a = dict(zd=32, xd=12, yd=12, xs=2, ys=2, xf=6, yf=6, nf=64)
    assert geometry.limb_exec_over_useful("conv", a, 128, 64 * 16) == 1.0
    # a dense layer of 800 -> 500 on 64 images: 25 steps -> 26 (zero step), 500 -> 512 filters
    assert abs(geometry.limb_exec_over_useful("fc", dict(in_dim=800, out_dim=500), 64, 500) - (26 * 32 / 800) * (512 / 500)) < 1e-12
