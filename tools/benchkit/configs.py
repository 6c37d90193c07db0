"""the workloads bench.py measures (BASELINE.json configs), the seeds of its client side, and the reference-made goldens they are checked against"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INT8_PEAK_TOPS = 5000.0        # dense int8 MFMA peak: 2x the ~2.5 PF bf16 rate per clock (MI355X_MICROARCH.md, matrix cores; no sparsity)

CONFIGS = {
    # BASELINE.json configs[1]: PlainModelTiny.h5, n=4096, batch=1024 on one MI355X  (q = coeff_modulus_128(4096), t = 2^20)
    "tiny4096": dict(model="PlainModelTiny", n=4096, k=2, t=1 << 32, batch=1024, chunk=128, distinct=32),   # t=2^32: exact logits without the client-side refresh
    # configs[2]: ApproxPlainModel.h5, n=8192, 3 coeff moduli, batch=1024
    "approx8192": dict(model="ApproxPlainModel", n=8192, k=3, t=1 << 42, batch=1024, chunk=32, tail=2, distinct=32),   # t=2^42: exact logits, 19 bits of budget left; dense layers per 64 images (+1 %)
    # configs[4]: PlainModelWoPad.h5, n=16384, 4 coeff moduli
    # (tail=5: the dense layers run once per 5 chunks = 30 images -- two-level chunking, netrun.prepare: fc3 streams 177 GiB of limb-form weights per launch and is
    # bound by its MFMAs, which a 64-row tile costs whether 48 (24 images, round 3) or 60 of its rows are real)
    "wopad16384": dict(model="PlainModelWoPad", n=16384, k=4, t=1 << 44, batch=1024, chunk=6, tail=5, distinct=12),
    # the coefficient modulus CrCNN itself would run at n=8192 (all four primes of coeff_modulus_128(8192)); at n=16384 the
    # eight default primes would need 424 GB for PlainModelWoPad's encoded weights alone (> HBM), so that one stays at k=4
    "approx8192k4": dict(model="ApproxPlainModel", n=8192, k=4, t=1 << 42, batch=1024, chunk=16, tail=2),      # (dense layers per 32 images: a full 64-row tile, +7 %)
    # every prime of coeff_modulus_128(16384), the coefficient modulus CrCNN's own setParameters(16384, t) picks: 424 GB of NTT-form weights -- fc3 keeps
    # coefficient-form plaintexts in HBM and is lifted + transformed a filter tile at a time inside every forward (netrun: streamed layers)
    "wopad16384k8": dict(model="PlainModelWoPad", n=16384, k=8, t=1 << 44, batch=96, chunk=4, tail=8),
    # ---- the reference's PUBLISHED configurations (Doc/Tesi.lyx), client-side refresh included (network.cpp:23,30-34): `reenc` = the layer
    # the refresh precedes, `published` = the thesis' seconds per image on its 40-core Xeon, in mainparams.cpp:81's column order
    "approx4096r": dict(model="ApproxPlainModel", n=4096, k=2, t=1 << 29, batch=512, chunk=64, tail=2, distinct=8, reenc=6, lossy=True,
                        published=dict(source="Doc/Tesi.lyx:12404,12492,13175-13708 (40-core Xeon E5-2640, one image)", total_s=69.07,
                                       columns=["T_LAYER_0", "T_LAYER_1", "T_LAYER_2", "T_LAYER_3", "T_LAYER_4", "T_LAYER_5", "T_REENC", "T_LAYER_6", "T_LAYER_7", "T_LAYER_8"],
                                       seconds=[30.73, 2.45, 2.03, 7.89, 0.65, 0.76, 3.20, 0.68, 18.23, 2.45])),
    "tiny2048r": dict(model="PlainModelTiny", n=2048, k=1, t=1 << 18, batch=1024, chunk=256, distinct=8, reenc=4, lossy=True,
                      published=dict(source="Doc/Tesi.lyx:14710,14789,15236-15586 (40-core Xeon E5-2640, one image, t = 2^18)", total_s=35.55,
                                     columns=["T_LAYER_0", "T_LAYER_1", "T_LAYER_2", "T_LAYER_3", "T_REENC", "T_LAYER_4", "T_LAYER_5"],
                                     seconds=[3.35, 1.22, 23.88, 0.39, 1.77, 4.34, 0.62])),
    # north_star's letter -- "no MFMA": the tiny4096 workload with Network::matrix_cores = false (mac3_kernel + row NTT), the path round 1 measured
    "tiny4096_valu": dict(model="PlainModelTiny", n=4096, k=2, t=1 << 32, batch=256, chunk=64, distinct=32, matrix_cores=0),
    # small ring for the tests of this script and single-GPU rehearsals of the multi-rank path (golden: net_tiny1024_eng.json)
    "tiny1024": dict(model="PlainModelTiny", n=1024, k=2, q=[0x7fffffff380001, 0x3fffffff000001], t=1 << 32, batch=48, chunk=24),
}


# golden fixtures (tests/golden/net_*.json, produced by the compiled reference: oracle/make_golden_nets.py) whose encrypted input is what
# this script generates for image 0 of rank 0 -- same parameter set, same seeded client side
GOLDEN_FOR = {"tiny2048r": "tiny2048r", "approx4096r": "approx4096r", "tiny4096_valu": "tiny4096_t32", "tiny4096": "tiny4096_t32", "approx8192": "approx8192_t42", "approx8192k4": "approx8192k4_t42", "wopad16384": "wopad16384_t44", "wopad16384k8": "wopad16384k8_t44", "tiny1024": "tiny1024_eng"}
KEY_SEED, EVK_SEED, ENC_SEED = 2024, 2025, 7000


def golden_check(cfg_name, cfg, q, rank, x0_sha, out0_sha, dec0_sha=None):
    """True / False when a reference-made golden exists for exactly these parameters and inputs, else None.  Goldens of networks with a client-side refresh pin the
    DECRYPTED outputs (dec_sha256: the re-encryption is randomised, the plaintexts behind it are not)"""
    path = os.path.join(ROOT, "tests", "golden", f"net_{GOLDEN_FOR.get(cfg_name, '')}.json")
    if rank != 0 or not os.path.exists(path):
        return None, None
    g = json.load(open(path))
    same = (g.get("input_gen") == "engine" and g["model"] == cfg["model"] and g["n"] == cfg["n"] and g["t"] == cfg["t"] and [int(v) for v in g["q"]] == [int(v) for v in q]
            and (g["key_seed"], g["evk_seed"], g["enc_seed"], g["image_index"]) == (KEY_SEED, EVK_SEED, ENC_SEED, 0))
    if not same:
        return None, None
    if "dec_sha256" in g:
        if g.get("layer_before_reenc") != cfg.get("reenc"):
            return None, None
        return bool(g["input_sha256"] == x0_sha and g["dec_sha256"] == dec0_sha), os.path.basename(path)
    return bool(g["input_sha256"] == x0_sha and g["out_sha256"] == out0_sha), os.path.basename(path)
