#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the compiled reference (oracle/_ref/ref_harness).  TEST INFRASTRUCTURE.

Runs only where /root/reference exists (the build container).  For each parameter set it
  1. creates keys / ciphertexts / plaintexts with the oracle's seeded client-side code,
  2. feeds them to the *reference itself* (SEAL 2.3.1 Evaluator and the CrCNN Layer classes, compiled in place),
  3. stores the reference's outputs (and the inputs, so a drifting oracle RNG is detected) as a compressed .npz.
tests/test_oracle_golden.py then replays the same inputs through the oracle and demands bit equality; the GPU
parity tests demand the same of the HIP path.

usage:  python oracle/make_golden.py [ops] [layers] [nets] [loader]
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

HARNESS = os.path.join(HERE, "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("CRC_REFERENCE", "/root/reference")

OP_SETS = {
    # name: (n, q list, t, number of cts, plaintext values)
    "ops_n256_k2_t20": (256, [0x7fffffff380001, 0x3fffffff000001], 1 << 20, 3, [0.25, -2.5, 1.0, 0.0202090591192245]),
    "ops_n256_k3_t30": (256, [0x7fffffff380001, 0x7ffffffef00001, 0x3fffffff000001], 1 << 30, 2, [-0.1307, 7.0, 0.3333]),
    "ops_n256_k1_q60_t30": (256, [0xffffffffffc0001], 1 << 30, 2, [0.5, -1.75]),   # aux base grows by one (baseconverter.cpp:47-56)
    "ops_n2048_k1_t18": (2048, [0x3fffffff000001], 1 << 18, 1, [0.4242, -3.0]),   # the committed driver's parameters (mainparams.cpp:75-76)
    # SEAL's small_mods_40bit (util/globals.cpp): 2^40 - c 2^s + 1 primes BELOW the folding reduction's validity bound (modarith.h
    # fold_constant): every 128-bit reduction of the Square pipeline must take the generic Barrett path here
    "ops_n256_k2_q40_t16": (256, [0xffffe80001, 0xffffc40001], 1 << 16, 2, [0.5, -1.75]),
    # plain modulus ABOVE the coefficient primes (t = 2^41 > q_i ~ 2^40, t < q): SEAL's !enable_fast_plain_lift branches (context.cpp:156-165,
    # evaluator.cpp:1447-1463) -- the boundary the second phase of CrCNN's plain-modulus search walks along (optimalParametersChooser.cpp:44-58)
    "ops_n256_k2_q40_t41": (256, [0xffffe80001, 0xffffc40001], 1 << 41, 2, [0.5, -1.75]),
}
FLOATS = [0.0, 1.0, -1.0, 0.25, -2.5, 0.5, -0.5, 1.5, 2.5, -0.4242129623889923, 2.8214867115020752, 100.125, -77.0,
          3.14159, 1e-9, -1e-7, 0.1307, 0.3081, 1.0 / 3.0, 12345.678, 0.020209059119224548, 4.656612873077393e-10]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run(mode, d):
    subprocess.check_call([HARNESS, mode, d])


def put(d, name, arr, dtype=np.uint64):
    np.ascontiguousarray(arr, dtype=dtype).tofile(os.path.join(d, name))


def get(d, name, shape=None):
    a = np.fromfile(os.path.join(d, name), dtype=np.uint64)
    return a.reshape(shape) if shape else a


def make_ops(name, n, q, t, nct, vals):
    O = orc.Oracle(n, q, t)
    k = O.k
    sk, pk = O.keygen(1000)
    evk = O.gen_evk(1001, sk)
    plains = np.zeros((len(vals), n), dtype=np.uint64)
    cc = np.zeros(len(vals), dtype=np.uint64)
    for j, v in enumerate(vals):
        plains[j], cc[j] = O.encode(np.float32(v) if abs(v) < 1e30 else v)
    msgs = O.encode_many([0.7071, -1.25, 3.5][:nct])
    cts = O.encrypt_many(pk, msgs, 2000)
    with tempfile.TemporaryDirectory() as d:
        put(d, "params.u64", [n, k, t] + list(q))
        put(d, "sk.u64", sk); put(d, "pk.u64", pk); put(d, "evk.u64", evk)
        put(d, "ct_in.u64", cts); put(d, "plains.u64", plains); put(d, "plain_cc.u64", cc)
        put(d, "floats.f64", FLOATS, dtype=np.float64)
        run("ops", d)
        npl = len(vals)
        g = dict(
            n=n, q=np.array(q, dtype=np.uint64), t=t, floats=np.array(FLOATS), plain_values=np.array(vals),
            sk=sk, pk=pk, evk=evk, ct_in=cts, plains=plains, plain_cc=cc, msgs=msgs,
            ref_consts=get(d, "ref_consts.u64"), ref_params_hash=get(d, "ref_params_hash.u64"),
            **({nm: np.fromfile(os.path.join(d, nm + ".bin"), dtype=np.uint8) for nm in ("ref_wire_ct", "ref_wire_evk", "ref_wire_pk", "ref_wire_sk")} if n <= 256 else {}),
            ref_root_powers0=get(d, "ref_root_powers0.u64", (2, n)),
            ref_dec_in=get(d, "ref_dec_in.u64", (nct, n)), ref_budget_in=get(d, "ref_budget_in.u64"),
            ref_enc=get(d, "ref_enc.u64", (npl, 2, k, n)),
            ref_sk=get(d, "ref_sk.u64", (k, n)), ref_pk=get(d, "ref_pk.u64", (2, k, n)), ref_evk=get(d, "ref_evk.u64"),
            ref_enc2=get(d, "ref_enc2.u64", (npl, 2, k, n)), ref_sq2=get(d, "ref_sq2.u64", (npl, 3, k, n)),
            ref_relin2=get(d, "ref_relin2.u64", (npl, 2, k, n)), ref_dec_relin2=get(d, "ref_dec_relin2.u64", (npl, n)),
            ref_ct_ntt=get(d, "ref_ct_ntt.u64", (nct, 2, k, n)), ref_plain_ntt=get(d, "ref_plain_ntt.u64", (npl, k, n)),
            ref_mul_ntt=get(d, "ref_mul_ntt.u64", (nct, npl, 2, k, n)), ref_mul=get(d, "ref_mul.u64", (nct, npl, 2, k, n)),
            ref_add=get(d, "ref_add.u64", (nct, 2, k, n)), ref_add_plain=get(d, "ref_add_plain.u64", (nct, npl, 2, k, n)),
            ref_sub_plain=get(d, "ref_sub_plain.u64", (nct, npl, 2, k, n)), ref_mul_plain=get(d, "ref_mul_plain.u64", (nct, npl, 2, k, n)),
            ref_sq=get(d, "ref_sq.u64", (nct, 3, k, n)), ref_relin=get(d, "ref_relin.u64", (nct, 2, k, n)),
            ref_budget_relin=get(d, "ref_budget_relin.u64"), ref_dec_relin=get(d, "ref_dec_relin.u64", (nct, n)),
            ref_enc_floats=get(d, "ref_enc_floats.u64", (len(FLOATS), n)), ref_enc_cc=get(d, "ref_enc_cc.u64"),
            ref_decode=get(d, "ref_decode.u64").view(np.float64),
        )
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **g)
    print("wrote", name, {k_: getattr(v, "shape", v) for k_, v in g.items() if k_.startswith("ref_")})


LAYER_SET = dict(n=256, q=[0x7fffffff380001, 0x3fffffff000001], t=1 << 20, zd=2, xd=5, yd=5,
                 conv=(2, 1, 3, 3, 3), fc_out=4, pool=(2, 1, 2, 2))


# a ONE-channel convolution (CrCNN's conv1 shape class: what the engine's CRC_NTTL1 matrix-core kernel takes), through the reference's own ConvolutionalLayer
LAYER_SET1 = dict(n=256, q=[0x7fffffff380001, 0x3fffffff000001], t=1 << 20, zd=1, xd=9, yd=10,
                  conv=(2, 1, 4, 3, 5), fc_out=4, pool=(2, 1, 2, 2))


def make_layers(S=None, out_name="layers_n256_k2_t20.npz", conv_only=False):
    S = S or LAYER_SET
    n, q, t = S["n"], S["q"], S["t"]
    O = orc.Oracle(n, q, t); k = O.k
    sk, pk = O.keygen(3000); evk = O.gen_evk(3001, sk)
    rng = np.random.RandomState(7)
    zd, xd, yd = S["zd"], S["xd"], S["yd"]
    img = rng.uniform(-1.5, 2.8, size=(zd, xd, yd)).astype(np.float32)
    x = O.encrypt_many(pk, O.encode_many(img).reshape(zd, xd, yd, n), 4000)
    xs, ys, xf, yf, nf = S["conv"]
    conv_w = rng.normal(0, 0.2, size=(nf, zd, xf, yf)).astype(np.float32); conv_b = rng.normal(0, 0.1, size=nf).astype(np.float32)
    fc_w = rng.normal(0, 0.2, size=(S["fc_out"], zd * xd * yd)).astype(np.float32); fc_b = rng.normal(0, 0.1, size=S["fc_out"]).astype(np.float32)
    bn_mean = rng.normal(0, 0.3, size=zd).astype(np.float32); bn_var = rng.uniform(0.5, 2.0, size=zd).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        put(d, "params.u64", [n, k, t] + list(q)); put(d, "evk.u64", evk)
        put(d, "layer_dims.u64", [zd, xd, yd, xs, ys, xf, yf, nf, S["fc_out"]] + list(S["pool"]) + [1 if conv_only else 0])
        put(d, "layer_in.u64", x)
        for nm, a in [("conv_w", conv_w), ("conv_b", conv_b), ("fc_w", fc_w), ("fc_b", fc_b), ("bn_mean", bn_mean), ("bn_var", bn_var)]:
            put(d, nm + ".f64", a.astype(np.float64), dtype=np.float64)
        run("layers", d)
        xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
        pxs, pys, pxf, pyf = S["pool"]
        pxo, pyo = (xd - pxf) // pxs + 1, (yd - pyf) // pys + 1
        g = dict(n=n, q=np.array(q, dtype=np.uint64), t=t, sk=sk, pk=pk, evk=evk, x=x, img=img,
                 dims=np.array([zd, xd, yd, xs, ys, xf, yf, nf, S["fc_out"]] + list(S["pool"])),
                 conv_w=conv_w, conv_b=conv_b, fc_w=fc_w, fc_b=fc_b, bn_mean=bn_mean, bn_var=bn_var,
                 ref_conv=get(d, "ref_conv.u64", (nf, xo, yo, 2, k, n)))
        if not conv_only:
            g.update(ref_fc=get(d, "ref_fc.u64", (1, S["fc_out"], 1, 2, k, n)),
                     ref_pool=get(d, "ref_pool.u64", (zd, pxo, pyo, 2, k, n)), ref_avgpool=get(d, "ref_avgpool.u64", (zd, pxo, pyo, 2, k, n)),
                     ref_bn=get(d, "ref_bn.u64", (zd, xd, yd, 2, k, n)), ref_square=get(d, "ref_square.u64", (zd, xd, yd, 2, k, n)))
    if conv_only:
        g = {k_: v for k_, v in g.items() if k_ in ("n", "q", "t", "x", "img", "dims", "conv_w", "conv_b", "ref_conv", "sk")}
    np.savez_compressed(os.path.join(GOLD, out_name), **g)
    print("wrote", out_name, {k_: v.shape for k_, v in g.items() if k_.startswith("ref_")})


# ---- model weights straight from the reference's .h5 files via the image's h5dump (pins the product's own HDF5 reader)
def h5_dataset(path, name):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.bin")
        subprocess.check_call(["/opt/conda/bin/h5dump", "-d", "/" + name, "-b", "LE", "-o", out, path], stdout=subprocess.DEVNULL)
        return np.fromfile(out, dtype="<f4")


def h5_names(path):
    txt = subprocess.check_output(["/opt/conda/bin/h5ls", path]).decode()
    return [ln.split()[0].replace("\\", "") for ln in txt.splitlines() if ln.strip()]


MODELS = ["PlainModelTiny", "ApproxPlainModel", "PlainModelWoPad"]


def make_loader():
    g = {}
    for m in MODELS:
        path = os.path.join(REF, "PlainModel", m + ".h5")
        for nm in h5_names(path):
            a = h5_dataset(path, nm)
            g[m + "/" + nm + "/sha256"] = sha(a)
            g[m + "/" + nm + "/count"] = a.size
            g[m + "/" + nm + "/head"] = a[:8].copy()
            g[m + "/" + nm + "/sum64"] = float(a.astype(np.float64).sum())
    np.savez_compressed(os.path.join(GOLD, "h5_datasets.npz"), **g)
    print("wrote loader fixture:", len(g) // 4, "datasets")


def make_files():
    """CrCNN's own file formats written BY THE REFERENCE (ref_harness files): an encoded-model stream (conv + batch-norm + dense parameters through
    savePlaintextParameters, cnnBuilder.cpp:181-196) and a cipher_image file (encryptAndSaveImage, globals.cpp:174-190), with the reference's outputs
    of those layers on that image.  tests/test_gpu_host_cpp.py has the C++ host classes load them, and the reference load what those write."""
    n, q, t = 256, [0x7fffffff380001, 0x3fffffff000001], 1 << 20
    O = orc.Oracle(n, q, t)
    sk, pk = O.keygen(4000)
    d = os.path.join(GOLD, "files_n256")
    os.makedirs(d, exist_ok=True)
    rng = np.random.RandomState(12)
    zd, xd, yd, xs, ys, xf, yf, nf, od = 1, 4, 4, 1, 1, 3, 3, 2, 3
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    put(d, "params.u64", [n, len(q), t] + q); put(d, "sk.u64", sk); put(d, "pk.u64", pk)
    put(d, "layer_dims.u64", [zd, xd, yd, xs, ys, xf, yf, nf, od])
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    put(d, "conv_w.f64", f32(rng.normal(0, 0.4, nf * zd * xf * yf)), dtype=np.float64); put(d, "conv_b.f64", f32(rng.normal(0, 0.2, nf)), dtype=np.float64)
    put(d, "bn_mean.f64", f32(rng.normal(0, 0.3, nf)), dtype=np.float64); put(d, "bn_var.f64", f32(rng.uniform(0.5, 2.0, nf)), dtype=np.float64)
    put(d, "fc_w.f64", f32(rng.normal(0, 0.3, od * nf * xo * yo)), dtype=np.float64); put(d, "fc_b.f64", f32(rng.normal(0, 0.1, od)), dtype=np.float64)
    put(d, "image.f64", f32(rng.uniform(-1, 1, zd * xd * yd)), dtype=np.float64)
    for stale in ("our_encoded_layers.bin", "our_cipher_image.bin"):
        if os.path.exists(os.path.join(d, stale)):
            os.remove(os.path.join(d, stale))
    run("files", d)
    print("wrote files_n256:", sorted(os.listdir(d)))


def search_cases():
    """(lo, hi, pow, first_good, last_good): the synthetic verdict tables of tests/test_search_logic.py, both phases of the reference's search"""
    q1 = 18014398492704769
    cases = []
    for lo_e, hi_e in [(16, 34), (24, 34), (20, 21), (20, 20), (10, 40), (1, 62)]:
        for fg_e in range(lo_e - 1, hi_e + 3, 3):
            for lg_e in (fg_e - 1, fg_e, fg_e + 2, hi_e + 1):
                cases.append((1 << lo_e, 1 << hi_e, 1, (1 << fg_e) + (fg_e % 2), (1 << max(lg_e, 0)) + 5))
    cases = cases[::3]
    # second phase (optimalParametersChooser.cpp:44-58): integers of [2^floor(log2 q), q - 1] for a small "smallest prime" q, and for the real one
    q = (1 << 20) + 7
    for fg, lg in [((1 << 20) + 3, 1 << 30), (1 << 20, 1 << 30), ((1 << 21) + 1, 1 << 30), (1 << 22, 1 << 21), ((1 << 20) + 6, 1 << 30), ((1 << 20) + 7, 1 << 30)]:
        cases.append((1 << 16, 1 << 34, 1, fg, lg))
        cases.append((1 << 20, q - 1, 0, fg, lg))
    cases.append((1 << 53, q1 - 1, 0, (1 << 53) + 12345, 1 << 60))
    return cases


def make_search():
    """candidate sequences of the REFERENCE's plainModulusBinarySearchInternal (oracle/_ref/search_harness) -> tests/golden/search_sequences.json"""
    import json
    harness = os.path.join(HERE, "_ref", "search_harness")
    if not os.path.exists(harness):
        subprocess.check_call(["make", "-C", HERE, "ref", "-j8"])
    out = []
    for lo, hi, pw, fg, lg in search_cases():
        # (linked against the image's libhdf5: _ref/h5libs holds symlinks to exactly those two libraries, oracle/Makefile)
        r = subprocess.run([harness, str(lo), str(hi), str(pw), str(fg), str(lg)], capture_output=True, text=True, check=True,
                           env=dict(os.environ, LD_LIBRARY_PATH=os.path.join(HERE, "_ref", "h5libs")))
        lines = r.stderr.splitlines()
        tried = [[int(l.split()[1]), l.split()[2]] for l in lines if l.startswith("tried")]
        found = int([l for l in lines if l.startswith("found")][0].split()[1])
        out.append(dict(lo=lo, hi=hi, pow=pw, first_good=fg, last_good=lg, tried=tried, found=found))
    json.dump(dict(source="CrCNN/src/optimalParametersChooser.cpp:84-181 (plainModulusBinarySearchInternal) compiled in place, predicate = table", cases=out),
              open(os.path.join(GOLD, "search_sequences.json"), "w"), indent=0)
    print("wrote search_sequences.json:", len(out), "cases")


if __name__ == "__main__":
    what = sys.argv[1:] or ["ops", "layers", "loader"]
    os.makedirs(GOLD, exist_ok=True)
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", HERE, "ref", "-j8"])
    if "ops" in what:
        for nm, (n, q, t, nct, vals) in OP_SETS.items():
            make_ops(nm, n, q, t, nct, vals)
    if "layers" in what:
        make_layers()
    if "layers1" in what:
        make_layers(LAYER_SET1, "layers1_n256_k2_t20.npz", conv_only=True)
    if "loader" in what:
        make_loader()
    if "files" in what:
        make_files()
    if "search" in what:
        make_search()
    if "nets" in what:
        from oracle import make_golden_nets
        make_golden_nets.main()
