"""the CPU baseline leg of bench.py: the compiled reference (oracle/_ref/ref_harness) or, where that binary is absent, the CPU oracle, timed on the node's host cores on a bounded sample"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def host_cores():
    """CPU cores this process may really use: scheduler affinity capped by the cgroup CPU quota (a GPU box gives one GPU's
    share of the host, not all of its hardware threads)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("CRC_CPU_THREADS", "16"))))


def cpu_baseline_reference(cfg, q, W, x0, cores, evk=None):
    """CPU baseline with the REFERENCE ITSELF: oracle/_ref/ref_harness (SEAL 2.3.1 + the CrCNN layer sources compiled in place by
    oracle/Makefile, shipped as a prebuilt binary) runs CrCNN's own ConvolutionalLayer::forward and pooling forward of the first
    two layers on image 0 with th_count = host cores; per-image time extrapolated by MAC count.  Returns None if the binary is absent."""
    import subprocess
    import tempfile
    from crcnn_amd.netrun import TOPOLOGIES, layer_macs
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return None
    topo = TOPOLOGIES[cfg["model"]]
    (k0, n0, a0), (k1, n1, a1) = topo[0], topo[1]
    with tempfile.TemporaryDirectory() as d:
        np.array([cfg["n"], len(q), cfg["t"]] + list(q), dtype=np.uint64).tofile(os.path.join(d, "params.u64"))
        np.array([1, 28, 28], dtype=np.uint64).tofile(os.path.join(d, "net_in_dims.u64"))
        np.ascontiguousarray(x0).tofile(os.path.join(d, "net_in.u64"))
        W[n0 + ".weight"].astype(np.float64).tofile(os.path.join(d, n0 + ".weight.f64")); W[n0 + ".bias"].astype(np.float64).tofile(os.path.join(d, n0 + ".bias.f64"))
        with open(os.path.join(d, "topology.txt"), "w") as f:
            f.write(f"conv {n0} {a0['xd']} {a0['yd']} {a0['zd']} {a0['xs']} {a0['ys']} {a0['xf']} {a0['yf']} {a0['nf']} {cores}\n")
            f.write(f"{k1} {n1} {a1['xd']} {a1['yd']} {a1['zd']} {a1['xs']} {a1['ys']} {a1['xf']} {a1['yf']}\n")
        try:
            subprocess.run([harness, "net", d], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            rows = [ln.split() for ln in open(os.path.join(d, "ref_digests.txt")).read().splitlines()]
        except Exception:
            return None
        # ... and, for the networks that have them, ONE ciphertext through CrCNN's BatchNormLayer and SquareLayer (Evaluator::square + relinearize): the first
        # ciphertext of the image as a 1 x 1 x 1 tensor
        t_bn = t_sq = None
        bn_l = [(n_, a_) for k_, n_, a_ in topo if k_ == "bn"]
        if bn_l and evk is not None and any(k_ == "square" for k_, _, _ in topo):
            try:
                d2 = os.path.join(d, "one"); os.makedirs(d2)
                np.array([cfg["n"], len(q), cfg["t"]] + list(q), dtype=np.uint64).tofile(os.path.join(d2, "params.u64"))
                np.array([1, 1, 1], dtype=np.uint64).tofile(os.path.join(d2, "net_in_dims.u64"))
                np.ascontiguousarray(x0.reshape(-1, 2, len(q), cfg["n"])[:1]).tofile(os.path.join(d2, "net_in.u64"))
                np.ascontiguousarray(evk).tofile(os.path.join(d2, "evk.u64"))
                bn_name = bn_l[0][0]
                W[bn_name + ".running_mean"][:1].astype(np.float64).tofile(os.path.join(d2, bn_name + ".running_mean.f64"))
                W[bn_name + ".running_var"][:1].astype(np.float64).tofile(os.path.join(d2, bn_name + ".running_var.f64"))
                with open(os.path.join(d2, "topology.txt"), "w") as f:
                    f.write(f"bn {bn_name} 1\nsquare act1 1\n")
                subprocess.run([harness, "net", d2], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                r2 = [ln.split() for ln in open(os.path.join(d2, "ref_digests.txt")).read().splitlines()]
                t_bn, t_sq = float(r2[0][4].rstrip("us")) * 1e-6, float(r2[1][4].rstrip("us")) * 1e-6
            except Exception:
                t_bn = t_sq = None
    t_conv, t_pool = float(rows[0][4].rstrip("us")) * 1e-6, float(rows[1][4].rstrip("us")) * 1e-6
    macs0 = layer_macs(k0, a0)
    total_macs = sum(layer_macs(k_, a_) for k_, _, a_ in topo)
    pool_cts = sum(a_["zd"] * ((a_["xd"] - a_["xf"]) // a_["xs"] + 1) * ((a_["yd"] - a_["yf"]) // a_["ys"] + 1) for k_, _, a_ in topo if k_ in ("pool", "avgpool"))
    cts1 = a1["zd"] * ((a1["xd"] - a1["xf"]) // a1["xs"] + 1) * ((a1["yd"] - a1["yf"]) // a1["ys"] + 1)
    t_image = total_macs / (macs0 / t_conv) + pool_cts * (t_pool / cts1)
    extra = "square/bn layers not sampled" if any(k_ in ("square", "bn") for k_, _, _ in topo) else "the network has no square / batch-norm layer"
    if t_bn is not None:
        # ciphertexts per image through each batch-norm / square layer (shapes follow the topology)
        shape, bn_cts, sq_cts = (1, 28, 28), 0, 0
        from crcnn_amd.netrun import out_shape
        for k_, _, a_ in topo:
            if k_ == "bn":
                bn_cts += int(np.prod(shape))
            elif k_ == "square":
                sq_cts += int(np.prod(shape))
            shape = out_shape(k_, a_, shape)
        t_image += bn_cts * t_bn + sq_cts * t_sq / cores        # (SquareLayer splits its ciphertexts over th_count threads; BatchNormLayer and the pools are single-threaded)
        extra = (f"one ciphertext through BatchNormLayer::forward in {t_bn * 1e3:.1f} ms (x {bn_cts} per image, single-threaded in the reference) and through SquareLayer::forward "
                 f"(Evaluator::square + relinearize, dbc 16) in {t_sq * 1e3:.1f} ms (x {sq_cts} per image / th_count={cores})")
    return dict(value=1.0 / t_image, unit="encrypted images/sec", cores=cores, kind="reference",
                sample=f"the compiled reference (SEAL 2.3.1 + CrCNN ConvolutionalLayer/{'Avg' if k1 == 'avgpool' else ''}PoolingLayer::forward, oracle/_ref/ref_harness) on image 0: "
                       f"{n0} {macs0} ct*pt MACs in {t_conv:.2f}s with th_count={cores}, {n1} in {t_pool:.2f}s (single-threaded in the reference); "
                       f"per-image time extrapolated by MAC count ({total_macs} MACs/image) and pooled-ciphertext count; {extra}",
                mac_per_s=macs0 / t_conv)


def cpu_baseline(cfg, q, W, x0, target_s):
    """the CPU oracle in the reference's operation order (per-product INTT, convolutionalLayer.cpp:73-88), th_count = host
    cores, timed on a bounded sample: conv1 restricted to as many filters as fit the time target, plus the first pooling
    layer; extrapolated to one image by MAC count (images and output channels are independent)."""
    from crcnn_amd.netrun import TOPOLOGIES, layer_macs
    from oracle import orc
    cores = host_cores()
    O = orc.Oracle(cfg["n"], q, cfg["t"])
    topo = TOPOLOGIES[cfg["model"]]
    kind, name, a = topo[0]
    enc = lambda v: O.encode_many(np.asarray(v, dtype=np.float32)).reshape(np.shape(v) + (O.n,))
    w = O.plains_to_ntt(enc(W[name + ".weight"].reshape(a["nf"], a["zd"], a["xf"], a["yf"])))
    b = enc(W[name + ".bias"])
    # calibrate on one filter with one thread, then size the sample
    t0 = time.time(); O.conv(x0, w, b, a["xs"], a["ys"], threads=1, f_range=(0, 1)); one = time.time() - t0
    macs_per_filter = layer_macs(kind, a) // a["nf"]
    nfil = int(max(1, min(a["nf"], (target_s * cores) // max(one, 1e-6))))
    nfil = max(min(nfil, a["nf"]), min(cores, a["nf"]))
    t0 = time.time(); y = O.conv(x0, w, b, a["xs"], a["ys"], threads=cores, f_range=(0, nfil)); t_conv = time.time() - t0
    mac_rate = nfil * macs_per_filter / t_conv
    total_macs = sum(layer_macs(k_, a_) for k_, _, a_ in topo)
    # pooling / bn / square: time the first pooling layer on the channels just computed, extrapolate by ciphertext count
    pk, pn, pa = topo[1]
    div = O.encode(1.0 / (pa["xf"] * pa["yf"]))[0] if pk == "avgpool" else None
    t0 = time.time(); O.pool(y[:nfil], pa["xs"], pa["ys"], pa["xf"], pa["yf"], div_plain=div, threads=cores); t_pool = time.time() - t0
    xo, yo = (pa["xd"] - pa["xf"]) // pa["xs"] + 1, (pa["yd"] - pa["yf"]) // pa["ys"] + 1
    pool_rate = nfil * xo * yo / max(t_pool, 1e-9)            # output cts per second (each: window adds + one multiply_plain)
    other_cts = 0
    for k_, _, a_ in topo:
        if k_ in ("pool", "avgpool"):
            other_cts += a_["zd"] * ((a_["xd"] - a_["xf"]) // a_["xs"] + 1) * ((a_["yd"] - a_["yf"]) // a_["ys"] + 1)
    t_image = total_macs / mac_rate + (other_cts / pool_rate if pk == "avgpool" else 0.0)
    return dict(value=1.0 / t_image, unit="encrypted images/sec", cores=cores, kind="port",
                sample=f"oracle (reference operation order) conv1 filters 0..{nfil - 1} of {a['nf']} + pool1 on image 0: "
                       f"{nfil * macs_per_filter} ct*pt MACs in {t_conv:.2f}s with {cores} threads; per-image time extrapolated by MAC count "
                       f"({total_macs} MACs/image){'' if pk == 'avgpool' else '; square/bn layers not sampled'}",
                mac_per_s=mac_rate)
