"""layer geometry behind bench.py's roofline: the fused layer plan, algorithmic bytes and multiply-adds, executed vs useful matrix-core work, offline PMC traffic"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def limb_exec_over_useful(kind, a, images, out_cts):
    """executed / useful int8 work of a layer on the limb GEMM (kernels_mfma.hip): rows = (image, pixel, poly) padded to 64-row tiles, the reduction to 32-term steps
    (an odd number of steps to even) -- per (tap, 32-channel block), or, for layers of fewer than 32 channels (the flat form), per 32-byte piece of a window row's
    (ky, channel) run with the channels rounded up to 4 -- and the filters to 32-filter tiles.  ApproxPlainModel's conv2 (20 channels, 3 x 3, 50 filters): 6 steps of
    32 for 180 terms since round 4 (was 10)."""
    zd, xf, yf, nf = (a["zd"], a["xf"], a["yf"], a["nf"]) if kind == "conv" else (a["in_dim"], 1, 1, a["out_dim"])
    rows = images * (out_cts // nf) * 2
    if zd < 32:
        zdc = -(-zd // 4) * 4
        ksteps = xf * -(-(yf * zdc) // 32)
    else:
        ksteps = -(-zd // 32) * xf * yf
    return (-(-rows // 64) * 64 / rows) * ((ksteps + (ksteps & 1)) * 32 / (zd * xf * yf)) * (-(-nf // 32) * 32 / nf)


def fused_plan(E, model, names=None):
    """the layer sequence Network::fuse() (crcnn_amd/host) leaves, with its geometry: [kind, name, geometry, input shape, output shape] per layer.  Pooling layers fold
    into the convolution in front of them where crc_plan_fold_pool says so (the cost model behind the C ABI), batch-norm layers into the conv / dense layer behind
    them.  Names are built the way the C++ classes build them, so bench.py can match bench_host's layer list against this plan.  `names` (bench_host's own list) settles
    the one choice the host makes from the HBM it finds: a batch norm is not folded into a layer whose weights are streamed (the fold is applied to NTT-form weights)."""
    from crcnn_amd.netrun import TOPOLOGIES, out_shape
    shape, plan = (1, 28, 28), []
    for kind, name, a in TOPOLOGIES[model]:
        o = out_shape(kind, a, shape)
        plan.append([kind, name, dict(a), shape, o]); shape = o
    folded, i = [], 0
    while i < len(plan):
        kind, name, a, ish, osh = plan[i]
        nxt = plan[i + 1] if i + 1 < len(plan) else None
        if kind == "conv" and nxt and nxt[0] in ("pool", "avgpool"):
            pa = nxt[2]
            if E.plan_fold_pool(a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], pa["xs"], pa["ys"], pa["xf"], pa["yf"]):
                a2 = dict(a, xf=(pa["xf"] - 1) * a["xs"] + a["xf"], yf=(pa["yf"] - 1) * a["ys"] + a["yf"], xs=a["xs"] * pa["xs"], ys=a["ys"] * pa["ys"])
                folded.append(["conv", name + "+" + nxt[1], a2, ish, nxt[4]]); i += 2
                continue
        folded.append(plan[i]); i += 1
    paired, i = [], 0
    while i < len(folded):        # Square + pooling -> one key switch per pooled ciphertext (SquarePoolLayer) where the engine's fp64 key switch holds the window's integers
        nxt = folded[i + 1] if i + 1 < len(folded) else None
        if folded[i][0] == "square" and nxt and nxt[0] in ("pool", "avgpool") and (E.square_pool_relin_supported(nxt[2]["xf"], nxt[2]["yf"]) if names is None else folded[i][1] + "+" + nxt[1] in names):
            paired.append(["squarepool", folded[i][1] + "+" + nxt[1], dict(nxt[2]), folded[i][3], nxt[4]]); i += 2
            continue
        paired.append(folded[i]); i += 1
    folded = paired
    out, i = [], 0
    while i < len(folded):
        nxt = folded[i + 1] if i + 1 < len(folded) else None
        if folded[i][0] == "bn" and nxt and nxt[0] in ("conv", "fc") and (names is None or folded[i][1] + "+" + nxt[1] in names):
            out.append([nxt[0], folded[i][1] + "+" + nxt[1], nxt[2], folded[i][3], nxt[4]]); i += 2
            continue
        out.append(folded[i]); i += 1
    return out


def layer_bytes_and_macs(E, kind, a, ishape, oshape, images):
    """algorithmic HBM bytes of one launch of a layer on `images` images -- every distinct operand moved once: 2 polys x k rows per input and output
    ciphertext, plus the layer's NTT-form weights -- and its ct x pt multiply-accumulates"""
    from crcnn_amd.netrun import layer_macs
    ct_bytes = 8 * E.k * E.n * 2
    wbytes = 0
    if kind == "conv":
        wbytes = a["nf"] * a["zd"] * a["xf"] * a["yf"] * 8 * E.k * E.n
    elif kind == "fc":
        wbytes = a["in_dim"] * a["out_dim"] * 8 * E.k * E.n
    return images * (int(np.prod(ishape)) + int(np.prod(oshape))) * ct_bytes + wbytes, layer_macs(kind, a) * images


def offline_traffic(cfg_name, kind, kernel_label, cts_per_launch):
    """HBM traffic of the dominant launch from the PMC counters: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes collected OFFLINE (tools/pmc_square.sh, tools/pmc_mac.sh) and
    committed under profiles/ -- bench.py cannot run the profiler on itself, so this is never measured in the run that quotes it.  Returns (bytes or None, source or None)."""
    for pf in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", pf))).get(cfg_name)
            if pm and pm.get("per_ciphertext") and kind == "square":      # the Square + relinearise sequence: PMC bytes per ciphertext x the launch's ciphertexts
                return int(pm["traffic_bytes_per_ciphertext"] * cts_per_launch), f"profiles/{pf} ({pm['kernel']}), offline rocprofv3 --pmc passes of tools/bench_square.py on the same ring (not measured in this run)"
            if pm and pm.get("pooled") and kind == "squarepool":          # the same with one key switch per pooled ciphertext: bytes per SQUARED ciphertext
                return int(pm["pooled"]["traffic_bytes_per_ciphertext"] * cts_per_launch), f"profiles/{pf} ({pm['pooled']['kernel']}), offline rocprofv3 --pmc passes of tools/bench_square_pool.py on the same ring (not measured in this run)"
            if pm and pm.get("kernel") == kernel_label:
                return int(pm["traffic_bytes"]), f"profiles/{pf}, offline rocprofv3 --pmc passes of the same launch (not measured in this run)"
        except Exception:
            pass
    return None, None
