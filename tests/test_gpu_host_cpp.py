"""GPU parity of the C++ host classes (crcnn_amd/host: Layer / Network / CnnBuilder with the reference's signatures):
the driver binary crcnn_amd/lib/test_host builds the real models with CnnBuilder from the HDF5 files and runs
Network::forward; per-layer outputs must hash to what the compiled reference produced (tests/golden/net_*.json)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from netcommon import GOLD, load_net_golden, make_inputs, sha

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "crcnn_amd", "lib", "test_host")


def run_driver(name, resident, batch=1, fuse=False, head_chunk=0, env=None, matrix_cores=True):
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    d = tempfile.mkdtemp()
    np.array([g["n"], len(g["q"]), g["t"]] + g["q"], dtype=np.uint64).tofile(os.path.join(d, "params.u64"))
    evk.tofile(os.path.join(d, "evk.u64")); x.tofile(os.path.join(d, "net_in.u64"))
    h5 = os.path.join(GOLD, "models", g["model"] + ".h5")
    subprocess.check_call([DRIVER, "net", g["model"], h5, d, "1" if resident else "0", str(batch), "1" if fuse else "0", str(head_chunk), "1" if matrix_cores else "0"], env=dict(os.environ, **(env or {})))
    return g, O, d


@pytest.mark.parametrize("name", ["tiny256", "approx256", "wopad256"])
def test_cpp_network_layerwise_digests(name):
    g, O, d = run_driver(name, resident=False)
    for i, L in enumerate(g["layers"]):
        t = np.fromfile(os.path.join(d, f"layer_{i}.u64"), dtype=np.uint64)
        assert sha(t) == L["sha256"], (name, i, L["name"])


@pytest.mark.parametrize("name", ["tiny256", "approx256", "wopad256"])
def test_cpp_network_resident_batch(name):
    g, O, d = run_driver(name, resident=True, batch=2)
    out = np.fromfile(os.path.join(d, "out.u64"), dtype=np.uint64).reshape(2, -1)
    assert sha(out[0]) == g["out_sha256"] and sha(out[1]) == g["out_sha256"]


def test_cpp_api_behaviour():
    """client-side round trip, resident == layerwise, refresh path, SEAL-format save/load, exceptions, HDF5 via CnnBuilder"""
    d = tempfile.mkdtemp()
    h5 = os.path.join(GOLD, "models", "PlainModelTiny.h5")
    out = subprocess.run([DRIVER, "api", h5, d], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "api ok" in out.stdout


def _broadcast_across_processes(world, env, devices):
    d = tempfile.mkdtemp()
    rdv = os.path.join(d, "rendezvous.id")
    procs = [subprocess.Popen([DRIVER, "bcast", str(r), str(world), rdv, d, str(devices[r])], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, **env)) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][1][-1500:])
        assert f"bcast ok: rank {r} of {world}" in outs[r][0]
    res = [np.fromfile(os.path.join(d, f"bcast_out_{r}.u64"), dtype=np.uint64) for r in range(world)]
    assert res[0].size > 0 and all(np.array_equal(res[0], x) for x in res[1:])


def test_cpp_broadcast_parameters_across_processes():
    """Network::broadcastParameters on a TWO-rank communicator, one process per rank: rank 1 builds the network from different weights, joins through the id rank 0
    left in a file, receives the encoded parameters and evaluation keys -- and then produces rank 0's output ciphertexts bit for bit.  On a one-GPU box the
    communicator's bytes go through the shared-memory rehearsal transport (RCCL refuses two ranks on one device); with two GPUs visible the same two processes
    also run over RCCL (ncclBroadcast)"""
    import torch
    _broadcast_across_processes(2, {"CRC_COMM_TRANSPORT": "shm"}, [0, 0])
    if torch.cuda.device_count() >= 2:
        _broadcast_across_processes(2, {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, [0, 1])


def test_cpp_plain_modulus_search():
    """the reference's plain-modulus binary search (optimalParametersChooser.cpp) driven by the GPU engine through
    the C++ host classes: setParameters / buildNetwork / encryptImage / Network::forward with budget check / decryptImage per
    candidate.  The run must be consistent with the reference's recursion replayed over the verdicts it observed, the modulus
    found must have succeeded, and the plaintext labels must be those of the float model."""
    from crcnn_amd import synth
    from oracle import search_ref as ref
    from benchkit.plain import plain_forward
    d = tempfile.mkdtemp()
    n, q = 4096, [0x7fffffff380001, 0x3fffffff000001]        # coeff_modulus_128(4096); the encoder needs this much room for a 6-layer net
    imgs = np.stack([synth.normalize(synth.synth_image(i)).reshape(-1) for i in range(4)]).astype(np.float32)
    imgs.tofile(os.path.join(d, "images.f32"))
    h5 = os.path.join(GOLD, "models", "PlainModelTiny.h5")
    lo, hi = 1 << 22, 1 << 26
    out = subprocess.check_output([DRIVER, "search", "PlainModelTiny", h5, os.path.join(d, "images.f32"), str(n), str(lo), str(hi), "2", "7"], text=True).split("\n")
    labels = [int(l.split()[2]) for l in out if l.startswith("label")]
    found = int([l for l in out if l.startswith("found")][0].split()[1])
    tried = [(int(l.split()[1]), l.split()[2]) for l in out if l.startswith("tried")]
    # plaintext labels = argmax of the float forward
    from crcnn_amd import binding
    W = {nm: binding.h5_read(h5, nm) for nm in binding.h5_list(h5)}
    assert labels == [int(np.argmax(plain_forward("PlainModelTiny", W, im.reshape(28, 28)))) for im in imgs]
    # replay the reference's control flow over the observed verdicts: same sequence of candidates, same result
    table = dict(tried)
    seen = []
    def pred(t):
        seen.append((t, table[t])); return table[t]
    assert ref.search(pred, lo, hi, min(q)) == found
    assert seen == tried
    assert found > 0 and table[found] == "SUCCESS"
    assert all(s != "SUCCESS" for t, s in tried if t < found)
    # the verdicts have the expected shape: too-small moduli mispredict, too-large ones run out of budget
    assert any(s == "MISPREDICTED" for t, s in tried if t < found) or found == lo


FULL = [n for n in ["tiny4096_t32", "approx8192_t42", "wopad16384_t44"] if os.path.exists(os.path.join(GOLD, f"net_{n}.json"))]


_FULL_RUNS = {}


def full_size_outputs(name):
    """the three NTT-resident runs of one full-size model from ONE built network (test_host net3): building it -- 10^5..10^6 plaintexts encoded on the CPU, lifted and
    transformed -- is most of a case's time, so the cases of a model share the driver process (round 3 built the network once per case: 130 s of the GPU suite)"""
    if name not in _FULL_RUNS:
        g = load_net_golden(name)
        O, sk, pk, evk, img, x = make_inputs(g)
        d = tempfile.mkdtemp()
        np.array([g["n"], len(g["q"]), g["t"]] + g["q"], dtype=np.uint64).tofile(os.path.join(d, "params.u64"))
        evk.tofile(os.path.join(d, "evk.u64")); x.tofile(os.path.join(d, "net_in.u64"))
        h5 = os.path.join(GOLD, "models", g["model"] + ".h5")
        batch = 6 if name.startswith("wopad16384") else 16      # 202 GiB of weights leave room for six images' activations (bench.py's chunk for this configuration)
        subprocess.check_call([DRIVER, "net3", g["model"], h5, d, str(batch)])
        outs = {}
        for key, fn, b in [("unfused", "out_unfused.u64", 1), ("fused", "out_fused.u64", 1), ("fused-batch16", "out_fused_batch.u64", batch)]:
            outs[key] = np.fromfile(os.path.join(d, fn), dtype=np.uint64).reshape(b, -1)
        import shutil
        shutil.rmtree(d, ignore_errors=True)
        _FULL_RUNS[name] = (g, outs)
    return _FULL_RUNS[name]


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("case", ["unfused", "fused", "fused-batch16"])
def test_cpp_network_full_size_equals_reference(name, case):
    """the drop-in itself (C++ Layer / Network / CnnBuilder) at the BASELINE ring sizes and the plain moduli bench.py runs at:
    CnnBuilder reads the real model, Network::forward (NTT-resident, before and after Network::fuse -- which here follows a forward, so it rebuilds the canonical weights
    from the plaintexts) must produce the compiled reference's output ciphertexts bit for bit.  Batch 16 gives a dense layer 32 rows = (image, poly): the C++ classes'
    dense limb path (a batch-1 dense layer stays on the vector-ALU kernel) and the limb hand-overs conv -> dense, dense -> dense are golden-checked at BASELINE sizes too"""
    g, outs = full_size_outputs(name)
    out = outs[case]
    for b in range(out.shape[0]):
        assert sha(out[b]) == g["out_sha256"], (name, case, b)


@pytest.mark.parametrize("name", ["tiny256", "approx256", "wopad256"])
def test_cpp_network_fused_equals_reference(name):
    """Network::fuse() (conv+pool folding, batch-norm folding) must leave the network's output ciphertexts bit-identical to the
    compiled reference's"""
    g, O, d = run_driver(name, resident=True, batch=2, fuse=True)
    out = np.fromfile(os.path.join(d, "out.u64"), dtype=np.uint64).reshape(2, -1)
    assert sha(out[0]) == g["out_sha256"] and sha(out[1]) == g["out_sha256"]


@pytest.mark.parametrize("name", ["wopad256", "tiny256"])
def test_cpp_two_level_chunking_and_tilewise_weights(name):
    """Network::head_chunk (the layers in front of the first dense layer on sub-batches, the dense layers once on the whole batch, every chunk packed straight into the
    dense layer's limb tensor) and FullyConnectedLayer's tile-wise limb weights with the batch-norm fold applied per tile (forced on the small ring: at n = 16384 the
    canonical and the limb copy of PlainModelWoPad's fc3 do not fit in HBM together) -- 16 images in chunks of 3 (a ragged last chunk), each one the reference's ciphertexts"""
    g, O, d = run_driver(name, resident=True, batch=16, fuse=True, head_chunk=3, env={"CRC_FORCE_TILEWISE": "1"})
    out = np.fromfile(os.path.join(d, "out.u64"), dtype=np.uint64).reshape(16, -1)
    assert all(sha(out[b]) == g["out_sha256"] for b in range(16))


@pytest.mark.parametrize("name", ["wopad256", "tiny256"])
def test_cpp_tilewise_layer_outside_the_matrix_core_plan(name):
    """a tile-wise dense layer has no canonical weights, so whoever reaches it first must build its limb tensor -- also a direct
    Layer::forward (the driver's layer-by-layer mode calls every layer itself, coefficient form at every boundary: the reference's own digests must come out) and a
    Network with matrix_cores = false (the plan would keep the layer off the limb GEMM; it runs there all the same, the only form its weights exist in)"""
    g, O, d = run_driver(name, resident=False, env={"CRC_FORCE_TILEWISE": "1"})
    for i, L in enumerate(g["layers"]):
        t = np.fromfile(os.path.join(d, f"layer_{i}.u64"), dtype=np.uint64)
        assert sha(t) == L["sha256"], (name, i, L["name"])
    g, O, d = run_driver(name, resident=True, batch=3, env={"CRC_FORCE_TILEWISE": "1"}, matrix_cores=False)
    out = np.fromfile(os.path.join(d, "out.u64"), dtype=np.uint64).reshape(3, -1)
    assert all(sha(out[b]) == g["out_sha256"] for b in range(3))


def test_cpp_example_driver():
    """crcnn_amd/host/example_main.cpp: the reference's timed test driver (mainparams.cpp:64-116) written against the host classes --
    encrypt, Network::forward (with and without Network::fuse), decrypt, arg-max equal to the float model's"""
    from crcnn_amd import synth
    d = tempfile.mkdtemp()
    imgs = np.stack([synth.normalize(synth.synth_image(i)).reshape(-1) for i in range(2)]).astype(np.float32)
    imgs.tofile(os.path.join(d, "images.f32"))
    exe = os.path.join(ROOT, "crcnn_amd", "lib", "example_main")
    h5 = os.path.join(GOLD, "models", "PlainModelTiny.h5")
    for fuse in ("0", "1"):
        out = subprocess.check_output([exe, "PlainModelTiny", h5, os.path.join(d, "images.f32"), "4096", str(1 << 32), "2", fuse], text=True)
        lines = [l for l in out.split("\n") if l.startswith("OUTPUT:")]
        assert len(lines) == 2 and all(l.endswith("Success") for l in lines), out
        assert "SUMMARY: 2 of 2" in out
    # the reference's T_REENC column (mainparams.cpp:81; the refresh of network.cpp:29-37): in front of the layer it precedes, and a positive time
    out = subprocess.check_output([exe, "PlainModelTiny", h5, os.path.join(d, "images.f32"), "4096", str(1 << 32), "1", "0", "4"], text=True)
    head = [l for l in out.split("\n") if l.startswith("INDEX_IMG")][0].split(",")
    assert head[1:8] == ["T_LAYER_0", "T_LAYER_1", "T_LAYER_2", "T_LAYER_3", "T_REENC", "T_LAYER_4", "T_LAYER_5"], head
    line = [l for l in out.split("\n") if l.startswith("OUTPUT:")][0]
    cols = line.split(",")
    assert line.endswith("Success") and float(cols[5]) > 0.0, line


def test_cpp_crcnn_files_interchange_with_the_reference():
    """CrCNN's own files (not just SEAL's object formats): the encoded-model stream -- savePlaintextParameters of consecutive layers, cnnBuilder.cpp:181-196 --
    and a cipher_image file (encryptAndSaveImage / loadEncryptedImage, globals.cpp:174-205).  tests/golden/files_n256 holds both as WRITTEN BY THE COMPILED
    REFERENCE (oracle/make_golden.py files) with the reference's layer outputs on them.  The C++ host classes must (a) load them and reproduce those outputs
    bit for bit, (b) write an encoded-model stream that is byte-identical to the reference's, and (c) write a cipher_image file the reference loads
    (checked with the prebuilt reference harness when it is present: it runs the same layers on OUR files and must land on OUR output bits)."""
    import shutil
    src = os.path.join(GOLD, "files_n256")
    d = tempfile.mkdtemp()
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), d)
    out = subprocess.run([DRIVER, "files", d], capture_output=True, text=True)
    assert out.returncode == 0 and "files ok" in out.stdout, out.stderr[-2000:]
    rd = lambda nm: np.fromfile(os.path.join(d, nm), dtype=np.uint64)
    assert np.array_equal(rd("out_from_ref_files.u64"), rd("ref_files_out_own.u64"))                              # (a)
    assert open(os.path.join(d, "our_encoded_layers.bin"), "rb").read() == open(os.path.join(d, "ref_encoded_layers.bin"), "rb").read()     # (b)
    assert os.path.getsize(os.path.join(d, "our_cipher_image.bin")) == os.path.getsize(os.path.join(d, "ref_cipher_image.bin"))
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):                                                                                     # (c)
        subprocess.check_call([harness, "files", d], stdout=subprocess.DEVNULL)
        assert np.array_equal(rd("ref_files_out_ours.u64"), rd("out_from_our_files.u64"))
        dec = np.fromfile(os.path.join(d, "ref_files_dec_ours.f64")); want = np.fromfile(os.path.join(src, "ref_files_dec_own.f64"))
        assert np.allclose(dec, want, atol=1e-4)          # same image, same layers, different encryption randomness
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("name", ["tiny256", "wopad256"])
def test_cpp_network_streamed_weights_equal_reference(name):
    """the C++ classes' fall-back for layers whose NTT-form weights exceed HBM (every prime of coeff_modulus_128(16384) with PlainModelWoPad: 424 GB): coefficient-form
    plaintexts resident, lift + NTT of a filter tile inside every forward.  Forced on a small ring through CRC_STREAM_SHARE; resident and fused runs stay bit-identical"""
    os.environ["CRC_STREAM_SHARE"] = "1e-9"
    try:
        for fuse in (False, True):
            g, O, d = run_driver(name, resident=True, batch=2, fuse=fuse)
            out = np.fromfile(os.path.join(d, "out.u64"), dtype=np.uint64).reshape(2, -1)
            assert sha(out[0]) == g["out_sha256"] and sha(out[1]) == g["out_sha256"]
        # ... and with 16 images = 32 rows per launch the streamed dense / multi-channel conv layers take the matrix-core route: 64-filter limb tiles built from
        # 8-filter canonical sub-tiles inside the forward (round 4; netrun.py's twin: tests/test_gpu_nets.py::test_streamed_weights_on_the_matrix_cores)
        g, O, d = run_driver(name, resident=True, batch=16, fuse=True)
        out = np.fromfile(os.path.join(d, "out.u64"), dtype=np.uint64).reshape(16, -1)
        assert all(sha(out[b]) == g["out_sha256"] for b in range(16))
    finally:
        del os.environ["CRC_STREAM_SHARE"]


PUBLISHED = [n for n in ["tiny2048r", "approx4096r"] if os.path.exists(os.path.join(GOLD, f"net_{n}.json"))]


def _run_refresh_config(name, batch, fuse, head_chunk):
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    d = tempfile.mkdtemp()
    np.array([g["n"], len(g["q"]), g["t"]] + g["q"], dtype=np.uint64).tofile(os.path.join(d, "params.u64"))
    evk.tofile(os.path.join(d, "evk.u64")); x.tofile(os.path.join(d, "net_in.u64")); sk.tofile(os.path.join(d, "sk.u64")); pk.tofile(os.path.join(d, "pk.u64"))
    h5 = os.path.join(GOLD, "models", g["model"] + ".h5")
    out = subprocess.run([DRIVER, "netr", g["model"], h5, d, str(batch), str(g["layer_before_reenc"]), "1" if fuse else "0", str(head_chunk)], capture_output=True,
                         text=True)
    assert out.returncode == 0 and "netr ok" in out.stdout, out.stderr[-2000:]
    return g, d


@pytest.mark.parametrize("name", PUBLISHED)
@pytest.mark.parametrize("case", ["unfused", "fused-batch", "fused-chunked"])
def test_cpp_published_configurations_with_refresh(name, case):
    """The configurations the reference PUBLISHES (Doc/Tesi.lyx:12404 ApproxPlainModel n = 4096, k = 2, t = 2^29, refresh in front of bn2 -- the committed
    network.cpp:23; :14701 PlainModelTiny n = 2048, k = 1, t = 2^18, refresh in front of fc3), Network::forward with the refresh on the device
    (crc_refresh_dev).  Against the compiled reference running its OWN decryptImage / encryptImage at the same layer (oracle/make_golden_nets.py):
    bits in front of the refresh (per-layer digests), the floats the client sees at the refresh (bit patterns), and -- the re-encryption being randomised --
    the DECRYPTED output plaintexts behind it, polynomial for polynomial, with the reference's remaining noise budget.  Fused, in a batch, and with the head
    layers chunked (the refresh then runs chunk by chunk in front of the first dense layer)."""
    g = load_net_golden(name)
    batch, fuse, chunk = {"unfused": (1, False, 0), "fused-batch": (5, True, 0), "fused-chunked": (5, True, 2)}[case]
    g, d = _run_refresh_config(name, batch, fuse, chunk)
    n = g["n"]
    if not fuse:
        for i in range(g["layer_before_reenc"]):
            assert sha(np.fromfile(os.path.join(d, f"pre_{i}.u64"), dtype=np.uint64)) == g["layers"][i]["sha256"], (name, i)
    want_fl = np.array(g["reenc_floats_bits"], dtype=np.uint32)
    fl = np.fromfile(os.path.join(d, "reenc_floats.f32"), dtype=np.uint32).reshape(batch, -1)
    assert all(np.array_equal(fl[b], want_fl) for b in range(batch))
    want_dec = np.load(os.path.join(GOLD, f"net_{name}_dec.npz"))["dec"]
    dec = np.fromfile(os.path.join(d, "dec.u64"), dtype=np.uint64).reshape(batch, 10, n)
    assert sha(want_dec) == g["dec_sha256"]
    assert all(np.array_equal(dec[b], want_dec) for b in range(batch))
    bud = np.fromfile(os.path.join(d, "budget.u64"), dtype=np.uint64).reshape(batch, 10)
    assert int(bud.min()) >= min(g["budget"]) - 2 and int(bud.max()) <= max(g["budget"]) + 2, (bud, g["budget"])
    import shutil
    shutil.rmtree(d, ignore_errors=True)
