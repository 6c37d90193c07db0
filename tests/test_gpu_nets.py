"""GPU parity, whole networks: the real CrCNN models (real weights from the HDF5 files, one synthetic encrypted image)
through the engine, per-layer SHA-256 against what the compiled reference's Network::forward produced
(tests/golden/net_*.json, oracle/make_golden_nets.py) -- at n=256 for all three topologies and at the BASELINE.json
parameter sets (Tiny n=4096 k=2, Approx n=8192 k=3, WoPad n=16384 k=4) when their goldens are present."""
import os

import numpy as np
import pytest

from netcommon import GOLD, load_net_golden, make_inputs, sha, sha_device

pytestmark = pytest.mark.gpu

# *_t32 / _t42 / _t44: the plain moduli bench.py runs at (exact logits without the client-side refresh); approx8192k4: all four primes of
# coeff_modulus_128(8192), the coefficient modulus CrCNN's own setParameters picks; tiny1024_eng: the small workload of the bench tests
NAMES = [n for n in ["tiny256", "approx256", "wopad256", "tiny1024_eng", "tiny4096", "approx8192", "wopad16384",
                     "tiny4096_t32", "approx8192_t42", "approx8192k4_t42", "wopad16384_t44"]
         if os.path.exists(os.path.join(GOLD, f"net_{n}.json"))]


def run_net(name, resident, batch=1, fuse=False):
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    E = ca.Engine(g["n"], g["q"], g["t"], device=0)
    d_evk = E.upload(evk)
    net = Network(E, g["model"], h5_path=os.path.join(GOLD, "models", g["model"] + ".h5"), resident=resident, d_evk=d_evk)
    if fuse:
        net.fuse()
    net.prepare(batch)
    xb = np.ascontiguousarray(np.repeat(x[None], batch, axis=0))
    d_x = E.upload(xb)
    digests = {}

    def timer(i, lname, kind, phase):
        if phase == 1 and not resident:      # coefficient form between layers: the reference's own layer boundary
            per_image = int(np.prod(net.plan[i][5])) * 2 * E.k * E.n * 8
            digests[i] = [sha_device(E, net.buf[net.slots[i]], per_image, b * per_image) for b in range(batch)]

    d_out = net.forward(d_x, batch, timer=timer)
    out = E.download(d_out, (batch, 1, 10, 1, 2, E.k, E.n))
    run_net.last_plan = [(pl[0], pl[1]) for pl in net.plan]
    E.close()
    return g, O, sk, out, digests


@pytest.mark.parametrize("name", NAMES)
def test_layerwise_digests_match_reference(name):
    """coefficient form in/out of every layer, exactly the reference's Layer::forward contract"""
    g, O, sk, out, digests = run_net(name, resident=False)
    for i, L in enumerate(g["layers"]):
        assert digests[i][0] == L["sha256"], (name, i, L["name"])
    assert sha(out[0]) == g["out_sha256"]


@pytest.mark.parametrize("name", NAMES)
def test_ntt_resident_network_matches_reference(name):
    """NTT-resident pipeline (one INTT at the end / around Square), batch of 2: same final ciphertext bits"""
    g, O, sk, out, _ = run_net(name, resident=True, batch=2)
    assert sha(out[0]) == g["out_sha256"] and sha(out[1]) == g["out_sha256"]
    if g["n"] >= 1024:       # real parameter sets: the decrypted logits are the reference's logits
        got = [O.decrypt_value(sk, out[0, 0, j, 0]) for j in range(10)]
        assert got == g["logits"]
        assert [O.noise_budget(sk, out[0, 0, j, 0]) for j in range(3)] == g["budget"][:3]


@pytest.mark.parametrize("name", ["tiny256", "wopad256", "tiny1024_eng"])
def test_streamed_weights_match_reference(name, monkeypatch):
    """layers whose NTT-form weights would not fit in HBM (PlainModelWoPad at n = 16384 with all eight primes: 424 GB) keep coefficient-form plaintexts and lift + NTT
    a filter tile at a time inside every forward (netrun.Network.stream_share).  Forced here on small rings: every conv / dense layer streams; same reference digests"""
    monkeypatch.setenv("CRC_STREAM_SHARE", "1e-9")
    g, O, sk, out, digests = run_net(name, resident=False)
    for i, L in enumerate(g["layers"]):
        assert digests[i][0] == L["sha256"], (name, i, L["name"])
    g, O, sk, out, _ = run_net(name, resident=True, batch=2)
    assert sha(out[0]) == g["out_sha256"] and sha(out[1]) == g["out_sha256"]


@pytest.mark.parametrize("name", ["tiny256", "wopad256"])
def test_streamed_weights_on_the_matrix_cores(name, monkeypatch):
    """a streamed dense / conv layer with at least 32 rows per launch builds 64-filter limb tiles from 8-filter canonical sub-tiles (crc_limb_pack_weights_tile) and runs
    the limb GEMM on them (PlainModelWoPad's fc3 with all eight primes at n = 16384); 16 images here, every one the compiled reference's ciphertexts"""
    monkeypatch.setenv("CRC_STREAM_SHARE", "1e-9")
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    E = ca.Engine(g["n"], g["q"], g["t"], device=0)
    net = Network(E, g["model"], h5_path=os.path.join(GOLD, "models", g["model"] + ".h5"), resident=True, d_evk=E.upload(evk))
    net.prepare(16)
    kern = {pl[1]: pl[3].get("stream_kernel") for pl in net.plan if pl[3].get("streamed")}
    assert kern["classifier.fc3"].startswith("mfma_mac2w_kernel") and kern["pool2_features.conv2"].startswith("mfma_mac2w_kernel"), kern
    assert kern["pool1_features.conv1"] == "mac3_kernel" and kern["classifier.fc4"] == "mac3_kernel"      # one channel; ten filters with only 32 rows (crc_plan_mac)
    out = E.download(net.forward(E.upload(np.ascontiguousarray(np.repeat(x[None], 16, axis=0))), 16), (16, 1, 10, 1, 2, E.k, E.n))
    E.close()
    assert all(sha(out[b]) == g["out_sha256"] for b in range(16))


@pytest.mark.parametrize("name", ["wopad256", "approx256"])
def test_tilewise_limb_weights_and_two_level_chunking(name, monkeypatch):
    """PlainModelWoPad's fc3 at n = 16384, k = 4 is 202 GiB in canonical NTT form and 177 GiB in limb form: the two cannot sit in HBM together, so the limb weights are
    built a filter tile at a time straight from the plaintexts, the batch-norm layer in front folded into every tile (netrun._build_tilewise).  And a dense layer
    streams all of its weights per launch, so the dense layers run once per GROUP of chunks (netrun.prepare(tail_group)).  Both forced here on a small ring: 8 chunks
    of 2 images, the dense layers once on all 16 -- every image the compiled reference's ciphertexts"""
    import crcnn_amd as ca
    from crcnn_amd import netrun
    monkeypatch.setattr(netrun.Network, "_needs_tilewise", lambda self, kind, a, count: kind == "fc" and a["in_dim"] == 800)
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    E = ca.Engine(g["n"], g["q"], g["t"], device=0)
    net = netrun.Network(E, g["model"], h5_path=os.path.join(GOLD, "models", g["model"] + ".h5"), resident=True, d_evk=E.upload(evk))
    fc3 = [pl for pl in net.plan if pl[1].endswith("classifier.fc3")][0]
    assert fc3[3]["tilewise"]["built"] is False and fc3[3]["w_form"] == ca.NTTL
    net.fuse()
    net.prepare(2, tail_group=8)
    fc3 = [pl for pl in net.plan if pl[1].endswith("classifier.fc3")][0]
    assert fc3[1] == "pool2_features.norm2+classifier.fc3" and fc3[3]["tilewise"]["built"] == "folded" and net.G == 8 and net.plan[net.split] is fc3
    fc4 = net.plan[-1]
    assert fc4[3]["w_form"] == ca.NTTP and fc3[3]["in_form"] == ca.NTTL              # ten filters, 32 rows: fc4 stays on the vector-ALU kernel; fc3's input is the group's limb tensor
    xb = E.upload(np.ascontiguousarray(np.repeat(x[None], 2, axis=0)))
    out = E.download(net.forward_group([xb] * 8, 2), (16, 1, 10, 1, 2, E.k, E.n))
    assert all(sha(out[b]) == g["out_sha256"] for b in range(16))
    # a shorter last group
    out = E.download(net.forward_group([xb] * 3, 2), (6, 1, 10, 1, 2, E.k, E.n))
    assert all(sha(out[b]) == g["out_sha256"] for b in range(6))
    E.close()


@pytest.mark.parametrize("name", [n for n in ["approx256", "wopad256", "approx8192_t42", "approx8192k4_t42", "wopad16384_t44"] if n in NAMES])
def test_square_and_pooling_share_one_key_switch(name):
    """Network.fuse() pairs the Square layer with the pooling behind it (crc_square_pool_relin_forms: the digits of a window's c2's are summed before ONE key switch):
    ApproxPlainModel's average pooling (the divisor multiplies the pooled ciphertexts) and PlainModelWoPad's sum pooling, three images -- the compiled reference's output
    ciphertexts, logits and noise budget"""
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    E = ca.Engine(g["n"], g["q"], g["t"], device=0)
    net = Network(E, g["model"], h5_path=os.path.join(GOLD, "models", g["model"] + ".h5"), resident=True, d_evk=E.upload(evk))
    net.fuse(); net.prepare(3)
    kinds = [pl[0] for pl in net.plan]
    assert "squarepool" in kinds and "square" not in kinds and [pl[1] for pl in net.plan if pl[0] == "squarepool"] == ["act1+pool2"], [pl[1] for pl in net.plan]
    out = E.download(net.forward(E.upload(np.ascontiguousarray(np.repeat(x[None], 3, axis=0))), 3), (3, 1, 10, 1, 2, E.k, E.n))
    E.close()
    assert all(sha(out[b]) == g["out_sha256"] for b in range(3))
    if g["n"] >= 1024:
        assert [O.decrypt_value(sk, out[2, 0, j, 0]) for j in range(10)] == g["logits"]


def test_kernel_choice_follows_the_rows_per_launch():
    """PlainModelWoPad at a chunk of 6 images: conv2 (6 x 2 x 25 rows) is a limb GEMM, fc3 (12 rows = a fifth of a 64-row tile: every slot's weights would be
    streamed for a handful of rows) stays on the vector-ALU kernel -- the guard counts PIXELS per image, one for a dense layer; at 16 images fc3 moves over; fc4's ten
    filters never do"""
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network
    g = load_net_golden("wopad256")
    O, sk, pk, evk, img, x = make_inputs(g)
    for B, dense_limb in ((6, False), (16, True)):
        E = ca.Engine(g["n"], g["q"], g["t"], device=0)
        net = Network(E, g["model"], h5_path=os.path.join(GOLD, "models", g["model"] + ".h5"), resident=True, d_evk=E.upload(evk))
        net.fuse(); net.prepare(B)
        forms = {pl[1].split("+")[-1]: pl[3] for pl in net.plan if pl[0] in ("conv", "fc")}
        assert forms["pool2_features.conv2"].get("w_form") == ca.NTTL
        assert (forms["classifier.fc3"].get("w_form") == ca.NTTL) == dense_limb, (B, forms["classifier.fc3"].get("w_form"))
        if not dense_limb:
            assert forms["classifier.fc3"]["limb_skipped"] == "fewer than 32 rows per launch"
        assert forms["classifier.fc4"].get("w_form") == ca.NTTP           # ten filters: a limb GEMM only from a full 64-row tile on (crc_plan_mac)
        out = E.download(net.forward(E.upload(np.ascontiguousarray(np.repeat(x[None], B, axis=0))), B), (B, 1, 10, 1, 2, E.k, E.n))
        assert all(sha(out[b]) == g["out_sha256"] for b in range(B))
        E.close()


def test_matrix_core_dense_layers_match_reference():
    """with at least half a 64-row tile per launch (16 images x 2 polys) the dense layers run on the matrix-core kernel too, and conv2 hands its tensor to fc3 in limb
    form (CRC_NTTL): every image of the batch must still come out as the compiled reference's ciphertexts"""
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network
    g = load_net_golden("tiny1024_eng")
    O, sk, pk, evk, img, x = make_inputs(g)
    E = ca.Engine(g["n"], g["q"], g["t"], device=0)
    net = Network(E, g["model"], h5_path=os.path.join(GOLD, "models", g["model"] + ".h5"), resident=True)
    net.fuse(); net.prepare(16)
    forms = {pl[1]: pl[3].get("w_form") for pl in net.plan if pl[0] in ("conv", "fc")}
    assert forms["classifier.fc3"] == ca.NTTL and forms["classifier.fc4"] == ca.NTTP and [pl[3]["in_form"] for pl in net.plan if pl[1] == "classifier.fc3"] == [ca.NTTL]
    d_x = E.upload(np.ascontiguousarray(np.repeat(x[None], 16, axis=0)))
    out = E.download(net.forward(d_x, 16), (16, 1, 10, 1, 2, E.k, E.n))
    assert all(sha(out[b]) == g["out_sha256"] for b in range(16))
    # a full tile of rows (32 images): the ten-filter fc4 moves to the matrix cores as well and fc3 hands its tensor over in limb form
    net2 = Network(E, g["model"], h5_path=os.path.join(GOLD, "models", g["model"] + ".h5"), resident=True)
    net2.fuse(); net2.prepare(32)
    f2 = {pl[1]: pl[3] for pl in net2.plan if pl[0] == "fc"}
    assert f2["classifier.fc4"]["w_form"] == ca.NTTL and f2["classifier.fc3"]["out_form"] == ca.NTTL
    out = E.download(net2.forward(E.upload(np.ascontiguousarray(np.repeat(x[None], 32, axis=0))), 32), (32, 1, 10, 1, 2, E.k, E.n))
    E.close()
    assert all(sha(out[b]) == g["out_sha256"] for b in range(32))


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "net_tiny4096_t32.json")), reason="needs the n=4096 golden")
def test_bench_chunk_matches_reference():
    """PlainModelTiny at the BASELINE parameters and a bench-sized chunk (32 images per layer launch; bench.py itself verifies its 128-image chunks against the same
    golden): conv1+pool1 on its own matrix-core kernel writing conv2's limb tensor, conv2 / fc3 / fc4 as limb GEMMs handing their tensors on in limb form -- every image
    of the chunk is the compiled reference's ciphertext"""
    import crcnn_amd as ca
    g, O, sk, out, _ = run_net("tiny4096_t32", resident=True, batch=32)
    bad = [b for b in range(32) if sha(out[b]) != g["out_sha256"]]
    assert not bad, bad


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "net_wopad16384k8_t44.json")), reason="needs the n=16384 k=8 golden")
def test_all_eight_primes_match_reference():
    """PlainModelWoPad at n = 16384 with all EIGHT primes of coeff_modulus_128(16384) -- the coefficient modulus CrCNN's own setParameters picks (globals.cpp) -- and
    t = 2^44 >= q_i / 2^11 (the slow plain lift).  fc3's NTT-form weights (419 GB) do not fit: the layer streams coefficient-form plaintexts (netrun stream_share).
    One image, NTT-resident: the compiled reference's ciphertext (an hour of CPU time in oracle/make_golden_nets.py).
    Then the same after Network.fuse(): Square + pooling with ONE key switch per pooled ciphertext (crc_square_pool_relin_forms) where its integers are largest --
    D = 32 digit polynomials, a 2 x 2 window: up to 2^91 of the 2^92.98 the two fp64 primes hold -- against the same reference ciphertext"""
    g, O, sk, out, _ = run_net("wopad16384k8_t44", resident=True, batch=1)
    assert sha(out[0]) == g["out_sha256"]
    g, O, sk, out, _ = run_net("wopad16384k8_t44", resident=True, batch=1, fuse=True)
    assert ("squarepool", "act1+pool2") in run_net.last_plan, run_net.last_plan
    assert sha(out[0]) == g["out_sha256"]


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "c1_tiny4096_t32.json")), reason="needs the configs[0] fixture")
def test_baseline_configs0_in_full():
    """BASELINE configs[0]: PlainModelTiny.h5 on 32 encrypted MNIST-like images at n = 4096.  The compiled reference ran all 32 (tests/golden/c1_tiny4096_t32.json,
    oracle/make_c1.py: 3.5 hours of CPU); the engine runs them as ONE chunk.  Same input ciphertexts (seeded client), same 10 output ciphertexts per image bit for bit,
    same predictions"""
    import hashlib
    import json
    import crcnn_amd as ca
    from crcnn_amd import synth
    from crcnn_amd.netrun import Network
    c1 = json.load(open(os.path.join(GOLD, "c1_tiny4096_t32.json")))
    assert len(c1["images"]) == 32
    E = ca.Engine(c1["n"], c1["q"], c1["t"], device=0)
    sk, pk = E.keygen(c1["key_seed"])
    xs = []
    for i in range(32):         # (crc_encrypt spreads an image's 784 ciphertexts over the host threads: one keystream per ciphertext)
        pl, _ = E.encode(synth.normalize(synth.synth_image(i)).reshape(-1))
        x = E.encrypt(pk, pl, c1["enc_seed_base"] + c1["enc_seed_stride"] * i)
        assert sha(x) == c1["images"][str(i)]["input_sha256"], i
        xs.append(x.reshape(28 * 28, 2, E.k, E.n))
    net = Network(E, "PlainModelTiny", h5_path=os.path.join(GOLD, "models", "PlainModelTiny.h5"), resident=True)
    net.prepare(32)
    out = E.download(net.forward(E.upload(np.ascontiguousarray(np.stack(xs))), 32), (32, 10, 2, E.k, E.n))
    bad = [i for i in range(32) if sha(out[i]) != c1["images"][str(i)]["out_sha256"]]
    assert not bad, bad
    from oracle import orc
    O = orc.Oracle(c1["n"], c1["q"], c1["t"])
    for i in (0, 13, 31):
        logits = [O.decrypt_value(sk, out[i, j]) for j in range(10)]
        assert int(np.argmax(logits)) == c1["images"][str(i)]["prediction"] and np.allclose(logits, c1["images"][str(i)]["logits"], atol=1e-9)
    E.close()
