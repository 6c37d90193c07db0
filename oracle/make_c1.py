#!/usr/bin/env python3
"""BASELINE configs[0] in full: PlainModelTiny.h5 on 32 encrypted synthetic MNIST-like images through the COMPILED REFERENCE
(oracle/_ref/ref_harness = SEAL 2.3.1 + CrCNN's Network::forward), n = 4096, q = coeff_modulus_128(4096), t = 2^32 (the plain modulus
bench.py runs at; exact logits without the client-side refresh).  TEST INFRASTRUCTURE; build container only (hours of CPU).

Keys and ciphertexts come from the engine's seeded host-side client exactly as bench.py makes them (key seed 2024, image i encrypted with
seed 7000 + 1000 i), so bench.py's first images are these images.  Writes tests/golden/c1_tiny4096_t32.json: per image the SHA-256 of
the 10 output ciphertexts, the decrypted logits, the prediction, the reference's per-layer seconds -- and the wall time of the whole
job, which is the CPU baseline of configs[0] measured rather than extrapolated.
   python oracle/make_c1.py [first] [count] [threads]        (resumable: images already in the JSON are skipped)"""
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
from oracle.make_golden_nets import topology, ENG_KEY_SEED, ENG_EVK_SEED, ENG_ENC_SEED  # noqa: E402

OUT = os.path.join(GOLD, "c1_tiny4096_t32.json")


def main(first=0, count=32, threads=8):
    import crcnn_amd as ca
    from crcnn_amd import synth
    from crcnn_amd.netrun import TOPOLOGIES
    model, n, q, t = "PlainModelTiny", 4096, orc.COEFF_MODULUS_128[4096], 1 << 32
    E = ca.Engine(n, q, t, device=-1)
    sk, pk = E.keygen(ENG_KEY_SEED); evk = E.gen_evk(ENG_EVK_SEED, sk)
    path = os.path.join(REF, "PlainModel", model + ".h5")
    res = json.load(open(OUT)) if os.path.exists(OUT) else dict(model=model, n=n, q=[int(v) for v in q], t=t, key_seed=ENG_KEY_SEED, enc_seed_base=ENG_ENC_SEED,
                                                                enc_seed_stride=1000, input_gen="engine", ref_threads=threads, images={})
    for idx in range(first, first + count):
        if str(idx) in res["images"]:
            continue
        img = synth.normalize(synth.synth_image(idx))
        pl, _ = E.encode(img.reshape(-1))
        x = E.encrypt(pk, pl, ENG_ENC_SEED + 1000 * idx).reshape(1, 28, 28, 2, len(q), n)
        t0 = time.time()
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            put(d, "params.u64", [n, len(q), t] + list(q)); put(d, "evk.u64", evk); put(d, "sk.u64", sk)
            put(d, "net_in_dims.u64", [1, 28, 28]); put(d, "net_in.u64", x)
            open(os.path.join(d, "topology.txt"), "w").write(topology(model, 2, th=threads))
            for kind, lname, a in TOPOLOGIES[model]:
                for suffix in {"conv": ["weight", "bias"], "fc": ["weight", "bias"]}.get(kind, []):
                    put(d, f"{lname}.{suffix}.f64", h5_dataset(path, f"{lname}.{suffix}").astype(np.float64), dtype=np.float64)
            subprocess.check_call([HARNESS, "net", d], stdout=subprocess.DEVNULL)
            rows = [ln.split() for ln in open(os.path.join(d, "ref_digests.txt")).read().splitlines()]
            out = get(d, "ref_net_out.u64", (1, 10, 1, 2, len(q), n))
            logits = [float(v) for v in get(d, "ref_net_logits.u64").view(np.float64)]
        res["images"][str(idx)] = dict(input_sha256=hashlib.sha256(x.tobytes()).hexdigest(), out_sha256=hashlib.sha256(out.tobytes()).hexdigest(), logits=logits,
                                       prediction=int(np.argmax(logits)), layer_seconds=[float(r[4].rstrip("us")) * 1e-6 for r in rows], wall_s=round(time.time() - t0, 1))
        done = res["images"]
        res["total_wall_s"] = round(sum(v["wall_s"] for v in done.values()), 1)
        res["images_per_s"] = round(len(done) / res["total_wall_s"], 6)
        json.dump(res, open(OUT, "w"), indent=1)
        print("image", idx, "prediction", res["images"][str(idx)]["prediction"], "wall", res["images"][str(idx)]["wall_s"], flush=True)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    main(*a)
