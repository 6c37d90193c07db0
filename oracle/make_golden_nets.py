#!/usr/bin/env python3
"""Whole-network goldens from the compiled reference: SHA-256 of every layer output for the real models on one
synthetic encrypted image, at the BASELINE parameter sets.  TEST INFRASTRUCTURE; runs only in the build container
(needs /root/reference, ~40 GB RAM and several CPU-minutes per model).

  python oracle/make_golden_nets.py [tiny4096] [approx8192] [wopad16384] [tiny256] ...
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from oracle.make_golden import GOLD, HARNESS, REF, h5_dataset, put, get  # noqa: E402

Q8192 = orc.COEFF_MODULUS_128[8192]
Q16384 = orc.COEFF_MODULUS_128[16384]
# name: (model, n, q, t, fc3 slices, reference thread counts)
NETS = {
    "tiny256": ("PlainModelTiny", 256, [0x7fffffff380001, 0x3fffffff000001], 1 << 20, 1),
    "approx256": ("ApproxPlainModel", 256, [0x7fffffff380001, 0x7ffffffef00001, 0x3fffffff000001], 1 << 30, 1),
    "wopad256": ("PlainModelWoPad", 256, [0x7fffffff380001, 0x7ffffffef00001, 0x3fffffff000001], 1 << 30, 1),
    "tiny4096": ("PlainModelTiny", 4096, orc.COEFF_MODULUS_128[4096], 1 << 20, 2),                      # BASELINE configs[0..1]
    "approx8192": ("ApproxPlainModel", 8192, Q8192[:3], 1 << 30, 10),                                   # configs[2..3]
    "wopad16384": ("PlainModelWoPad", 16384, Q16384[:4], 1 << 30, 25),                                  # configs[4]
    # the plain moduli bench.py actually runs (exact logits without the client-side refresh), and the
    # coefficient modulus CrCNN's own setParameters would pick at n=8192 (all four primes of coeff_modulus_128(8192))
    "tiny4096_t32": ("PlainModelTiny", 4096, orc.COEFF_MODULUS_128[4096], 1 << 32, 2),
    "approx8192_t42": ("ApproxPlainModel", 8192, Q8192[:3], 1 << 42, 10),
    "approx8192k4_t42": ("ApproxPlainModel", 8192, Q8192[:4], 1 << 42, 10),
    "wopad16384_t44": ("PlainModelWoPad", 16384, Q16384[:4], 1 << 44, 25),
    # all eight primes of coeff_modulus_128(16384): what CrCNN's own setParameters(16384, t) picks.  424 GB of NTT-form weights: the engine streams them
    "wopad16384k8_t44": ("PlainModelWoPad", 16384, Q16384[:8], 1 << 44, 50),
}
# The reference's PUBLISHED configurations (Doc/Tesi.lyx:12404,12492 and :14701,15468), client-side refresh included: the committed
# Network::forward refreshes in front of layer 6 (network.cpp:23) = bn2 of ApproxPlainModel; the thesis' PlainModelTiny run refreshes in front of fc3 (layer 4).
# Digests of the layers in front of the refresh are bits; behind it the re-encryption is randomised, so the goldens hold the floats the client saw at the refresh
# and the DECRYPTED outputs (plaintext polynomials + logits), which do not depend on the fresh noise.
NETS["tiny2048r"] = ("PlainModelTiny", 2048, orc.COEFF_MODULUS_128[2048], 1 << 18, 1)
NETS["approx4096r"] = ("ApproxPlainModel", 4096, orc.COEFF_MODULUS_128[4096], 1 << 29, 4)
REENC = {"tiny2048r": 4, "approx4096r": 6}
KEY_SEED, EVK_SEED, ENC_SEED, IMAGE_INDEX = 9000, 9001, 100000, 0
# The bench-parameter sets take their keys and encrypted input image from the ENGINE's seeded client side (crc_keygen / crc_gen_evk /
# crc_encrypt on the host, no GPU) with exactly the seeds bench.py uses for image 0 of rank 0: the product can then reproduce the golden
# input without touching anything under oracle/, and bench.py compares its output digest with the reference's (`golden_match`).
ENGINE_INPUTS = {"tiny4096_t32", "approx8192_t42", "approx8192k4_t42", "wopad16384_t44", "wopad16384k8_t44", "tiny1024_eng", "tiny2048r", "approx4096r"}
ENG_KEY_SEED, ENG_EVK_SEED, ENG_ENC_SEED = 2024, 2025, 7000
NETS["tiny1024_eng"] = ("PlainModelTiny", 1024, [0x7fffffff380001, 0x3fffffff000001], 1 << 32, 1)     # n = 1024: smallest ring in which the fractional encoding survives the four multiplicative levels


def topology(model, slices, th=int(os.environ.get("CRC_REF_THREADS", "8"))):
    """reference topology lines (cnnBuilder.cpp:115-169) for ref_harness; thread counts only affect speed"""
    from crcnn_amd.netrun import TOPOLOGIES
    lines = []
    for kind, name, a in TOPOLOGIES[model]:
        if kind == "conv":
            lines.append(f"conv {name} {a['xd']} {a['yd']} {a['zd']} {a['xs']} {a['ys']} {a['xf']} {a['yf']} {a['nf']} {th}")
        elif kind == "fc":
            if a["in_dim"] * a["out_dim"] > 100000 and slices > 1:
                lines.append(f"fcs {name} {a['in_dim']} {a['out_dim']} {th} {slices}")
            else:
                lines.append(f"fc {name} {a['in_dim']} {a['out_dim']} {th}")
        elif kind in ("pool", "avgpool"):
            lines.append(f"{kind} {name} {a['xd']} {a['yd']} {a['zd']} {a['xs']} {a['ys']} {a['xf']} {a['yf']}")
        elif kind == "bn":
            lines.append(f"bn {name} {a['ch']}")
        elif kind == "square":
            lines.append(f"square {name} {th}")
    return "\n".join(lines) + "\n"


def net_input(O, pk):
    img = orc.normalize(orc.synth_image(IMAGE_INDEX))
    return img, O.encrypt_many(pk, O.encode_many(img).reshape(1, 28, 28, O.n), ENC_SEED)


def make(name):
    model, n, q, t, slices = NETS[name]
    O = orc.Oracle(n, q, t)
    seeds = dict(key_seed=KEY_SEED, evk_seed=EVK_SEED, enc_seed=ENC_SEED, input_gen="oracle")
    if name in ENGINE_INPUTS:
        import crcnn_amd as ca
        from crcnn_amd import synth
        E = ca.Engine(n, q, t, device=-1)
        sk, pk = E.keygen(ENG_KEY_SEED); evk = E.gen_evk(ENG_EVK_SEED, sk)
        img = synth.normalize(synth.synth_image(IMAGE_INDEX))
        pl, _ = E.encode(img.reshape(-1))
        x = E.encrypt(pk, pl, ENG_ENC_SEED).reshape(1, 28, 28, 2, O.k, n)
        seeds = dict(key_seed=ENG_KEY_SEED, evk_seed=ENG_EVK_SEED, enc_seed=ENG_ENC_SEED, input_gen="engine")
    else:
        sk, pk = O.keygen(KEY_SEED); evk = O.gen_evk(EVK_SEED, sk)
        img, x = net_input(O, pk)
    path = os.path.join(REF, "PlainModel", model + ".h5")
    from crcnn_amd.netrun import TOPOLOGIES
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        put(d, "params.u64", [n, O.k, t] + list(q)); put(d, "evk.u64", evk); put(d, "sk.u64", sk)
        put(d, "net_in_dims.u64", [1, 28, 28]); put(d, "net_in.u64", x)
        if name in REENC:
            put(d, "reenc.u64", [REENC[name]]); put(d, "pk.u64", pk)
        open(os.path.join(d, "topology.txt"), "w").write(topology(model, slices))
        for kind, lname, a in TOPOLOGIES[model]:
            for suffix in {"conv": ["weight", "bias"], "fc": ["weight", "bias"], "bn": ["running_mean", "running_var"]}.get(kind, []):
                put(d, f"{lname}.{suffix}.f64", h5_dataset(path, f"{lname}.{suffix}").astype(np.float64), dtype=np.float64)
        subprocess.check_call([HARNESS, "net", d])
        digests = [ln.split() for ln in open(os.path.join(d, "ref_digests.txt")).read().splitlines()]
        out = get(d, "ref_net_out.u64", (1, 10, 1, 2, O.k, n))
        g = dict(model=model, n=n, q=[int(v) for v in q], t=t, **seeds, image_index=IMAGE_INDEX,
                 input_sha256=hashlib.sha256(x.tobytes()).hexdigest(),
                 layers=[dict(index=int(r[0]), name=r[1], shape=r[2], sha256=r[3], ref_time=r[4]) for r in digests],
                 logits=[float(v) for v in get(d, "ref_net_logits.u64").view(np.float64)],
                 budget=[int(v) for v in get(d, "ref_net_budget.u64")],
                 out_sha256=hashlib.sha256(out.tobytes()).hexdigest(), ref_wall_s=round(time.time() - t0, 1), ref_threads=8)
        if name in REENC:
            fl = get(d, "ref_reenc_floats.u64").astype(np.uint32).view(np.float32)
            dec = get(d, "ref_net_dec.u64", (10, n))
            g.update(layer_before_reenc=REENC[name], reenc_floats_bits=[int(v) for v in fl.view(np.uint32)],
                     reenc_floats_sha256=hashlib.sha256(fl.tobytes()).hexdigest(), ref_reenc_s=float(get(d, "ref_reenc_us.u64")[0]) / 1e6,
                     dec_sha256=hashlib.sha256(dec.tobytes()).hexdigest())
            del g["out_sha256"]                    # depends on the reference's own re-encryption randomness
            np.savez_compressed(os.path.join(GOLD, f"net_{name}_dec.npz"), dec=dec)
    json.dump(g, open(os.path.join(GOLD, f"net_{name}.json"), "w"), indent=1)
    if n <= 256:
        np.savez_compressed(os.path.join(GOLD, f"net_{name}_out.npz"), out=out)
    print("wrote", name, "wall", g["ref_wall_s"], "s; logits", np.round(g["logits"], 3), "budget", g["budget"][:3])


def main(which=None):
    for nm in (which or ["tiny256", "approx256", "wopad256"]):
        make(nm)


if __name__ == "__main__":
    main(sys.argv[1:] or None)
