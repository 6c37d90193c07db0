"""the CPU baseline leg of bench.py: the compiled reference (oracle/_ref/ref_harness) or, where that binary is absent, the CPU oracle, timed on the node's host cores on a bounded sample"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def host_cores(ranks_on_node=1):
    """CPU cores this process may really use: scheduler affinity capped by the cgroup CPU quota (a GPU box gives one GPU's
    share of the host, not all of its hardware threads), shared evenly by the ranks of the node, at most 16 (CRC_CPU_THREADS) per rank -- the same split
    crc_host::thread_limit makes for the C++ side: eight ranks on a 128-thread host get 16 each, on a 16-thread one 2 each"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n // max(1, ranks_on_node), int(os.environ.get("CRC_CPU_THREADS", "16"))))


def _ref_net(harness, d, cfg, q, dims, cts, topo_lines, floats, evk=None, timeout=900):
    """one `ref_harness net` run in directory d: parameters, input tensor `cts` of shape dims = [zd, xd, yd], topology lines, float parameter files.  Returns the rows of
    ref_digests.txt ([index, name, shape, sha256, '<microseconds>us']) or None"""
    import subprocess
    os.makedirs(d, exist_ok=True)
    np.array([cfg["n"], len(q), cfg["t"]] + list(q), dtype=np.uint64).tofile(os.path.join(d, "params.u64"))
    np.array(dims, dtype=np.uint64).tofile(os.path.join(d, "net_in_dims.u64"))
    np.ascontiguousarray(cts).tofile(os.path.join(d, "net_in.u64"))
    if evk is not None:
        np.ascontiguousarray(evk).tofile(os.path.join(d, "evk.u64"))
    for nm, arr in floats.items():
        np.asarray(arr, dtype=np.float64).tofile(os.path.join(d, nm + ".f64"))
    open(os.path.join(d, "topology.txt"), "w").write("\n".join(topo_lines) + "\n")
    try:
        subprocess.run([harness, "net", d], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
        return [ln.split() for ln in open(os.path.join(d, "ref_digests.txt")).read().splitlines()]
    except Exception:
        return None


def _us(row):
    return float(row[4].rstrip("us")) * 1e-6


def cpu_baseline_reference(cfg, q, W, x0, cores, evk=None, full_image=False):
    """CPU baseline with the REFERENCE ITSELF: oracle/_ref/ref_harness (SEAL 2.3.1 + the CrCNN layer sources compiled in place by oracle/Makefile, shipped as a prebuilt
    binary) on the node's host cores with th_count = cores.  Bounded sample, one piece per layer KIND, each extrapolated by its own multiply-accumulate count (round 5
    extrapolated the whole network from the one-channel conv1 and came out 7x faster than configs[0] measured in full: a product in a 32-channel layer costs the same, but
    the reference's per-output add_many over hundreds of terms and its thread split do not):
      conv1       the first min(nf, cores) filters of the real first layer on image 0 (+ the first pooling layer on their outputs)
      conv-shaped the first min(nf, cores) filters of the SECOND convolution on a window that yields one row of two output positions (its real channel count and taps)
      dense       the first min(out, cores) rows of the first dense layer (its real input width)
      bn / square one ciphertext each
    full_image: instead of sampling, the whole network on image 0 (minutes to an hour: --cpu-seconds >= 600).  Returns None if the binary is absent."""
    import tempfile
    from crcnn_amd.netrun import TOPOLOGIES, layer_macs, out_shape
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return None
    topo = TOPOLOGIES[cfg["model"]]
    k, n = len(q), cfg["n"]
    cts = np.ascontiguousarray(x0).reshape(-1, 2, k, n)                  # the 784 ciphertexts of image 0
    tile = lambda count: cts[np.arange(count) % len(cts)]

    def line(kind, name, a, nf=None, th=cores):
        if kind == "conv":
            return f"conv {name} {a['xd']} {a['yd']} {a['zd']} {a['xs']} {a['ys']} {a['xf']} {a['yf']} {nf or a['nf']} {th}"
        if kind == "fc":
            return f"fc {name} {a['in_dim']} {nf or a['out_dim']} {th}"
        if kind in ("pool", "avgpool"):
            return f"{kind} {name} {a['xd']} {a['yd']} {a['zd']} {a['xs']} {a['ys']} {a['xf']} {a['yf']}"
        if kind == "bn":
            return f"bn {name} {a['ch']}"
        return f"square {name} {th}"

    def params(kind, name, a, nf=None):
        if kind == "conv":
            return {name + ".weight": W[name + ".weight"].reshape(a["nf"], -1)[:nf or a["nf"]], name + ".bias": W[name + ".bias"][:nf or a["nf"]]}
        if kind == "fc":
            return {name + ".weight": W[name + ".weight"].reshape(a["out_dim"], -1)[:nf or a["out_dim"]], name + ".bias": W[name + ".bias"][:nf or a["out_dim"]]}
        if kind == "bn":
            return {name + ".running_mean": W[name + ".running_mean"], name + ".running_var": W[name + ".running_var"]}
        return {}

    with tempfile.TemporaryDirectory() as d:
        if full_image:
            lines, floats = [], {}
            for kind, name, a in topo:
                lines.append(line(kind, name, a)); floats.update(params(kind, name, a))
            rows = _ref_net(harness, os.path.join(d, "full"), cfg, q, [1, 28, 28], cts, lines, floats, evk=evk, timeout=7200)
            if rows is None:
                return None
            t_layers = [_us(r_) for r_ in rows]
            t_image = sum(t_layers)
            total_macs = sum(layer_macs(k_, a_) for k_, _, a_ in topo)
            return dict(value=1.0 / t_image, unit="encrypted images/sec", cores=cores, kind="reference",
                        sample=f"the compiled reference (oracle/_ref/ref_harness: SEAL 2.3.1 + CrCNN's layer classes) on ONE WHOLE image, th_count={cores}: {t_image:.1f} s; "
                               f"seconds per layer {[round(v, 2) for v in t_layers]}", mac_per_s=total_macs / t_image, measured_in_full=True)
        # ---- conv1 (+ pool1)
        (k0, n0, a0), (k1, n1, a1) = topo[0], topo[1]
        nf0 = min(a0["nf"], cores)
        a1s = dict(a1, zd=nf0)
        rows = _ref_net(harness, os.path.join(d, "c1"), cfg, q, [1, 28, 28], cts, [line(k0, n0, a0, nf0), line(k1, n1, a1s)], params(k0, n0, a0, nf0))
        if rows is None:
            return None
        t_conv, t_pool = _us(rows[0]), _us(rows[1])
        macs0 = layer_macs(k0, dict(a0, nf=nf0))
        rate = {"conv1": macs0 / t_conv}
        pooled_per_s = nf0 * ((a1["xd"] - a1["xf"]) // a1["xs"] + 1) * ((a1["yd"] - a1["yf"]) // a1["ys"] + 1) / t_pool
        notes = [f"{n0}: {nf0} of {a0['nf']} filters = {macs0} ct*pt MACs in {t_conv:.2f}s with th_count={cores}; {n1} on them in {t_pool:.2f}s (single-threaded in the reference)"]
        # ---- a convolution of the second layer's shape, on a window that gives one row of two outputs
        convs = [(k_, n_, a_) for k_, n_, a_ in topo if k_ == "conv"]
        if len(convs) > 1:
            _, n2, a2 = convs[1]
            nf2 = min(a2["nf"], cores)
            a2s = dict(a2, xd=a2["xf"], yd=a2["yf"] + a2["ys"])
            rows = _ref_net(harness, os.path.join(d, "c2"), cfg, q, [a2["zd"], a2s["xd"], a2s["yd"]], tile(a2["zd"] * a2s["xd"] * a2s["yd"]), [line("conv", n2, a2s, nf2)],
                            params("conv", n2, a2, nf2))
            if rows is not None:
                m2 = layer_macs("conv", dict(a2s, nf=nf2))
                rate["conv"] = m2 / _us(rows[0])
                notes.append(f"{n2}: {nf2} of {a2['nf']} filters x {a2['zd']} channels on a {a2s['xd']} x {a2s['yd']} window (two outputs each) = {m2} MACs in {_us(rows[0]):.2f}s")
        # ---- rows of the first dense layer
        fcs = [(k_, n_, a_) for k_, n_, a_ in topo if k_ == "fc"]
        if fcs:
            _, n3, a3 = fcs[0]
            r3 = min(a3["out_dim"], cores)
            rows = _ref_net(harness, os.path.join(d, "fc"), cfg, q, [1, a3["in_dim"], 1], tile(a3["in_dim"]), [line("fc", n3, a3, r3)], params("fc", n3, a3, r3))
            if rows is not None:
                rate["fc"] = r3 * a3["in_dim"] / _us(rows[0])
                notes.append(f"{n3}: {r3} of {a3['out_dim']} rows x {a3['in_dim']} inputs = {r3 * a3['in_dim']} MACs in {_us(rows[0]):.2f}s")
        # ---- one ciphertext through BatchNormLayer and SquareLayer
        t_bn = t_sq = None
        bn_l = [(n_, a_) for k_, n_, a_ in topo if k_ == "bn"]
        if bn_l and evk is not None and any(k_ == "square" for k_, _, _ in topo):
            bn_name = bn_l[0][0]
            rows = _ref_net(harness, os.path.join(d, "one"), cfg, q, [1, 1, 1], cts[:1], [f"bn {bn_name} 1", "square act1 1"],
                            {bn_name + ".running_mean": W[bn_name + ".running_mean"][:1], bn_name + ".running_var": W[bn_name + ".running_var"][:1]}, evk=evk)
            if rows is not None:
                t_bn, t_sq = _us(rows[0]), _us(rows[1])
    # ---- one image: every layer by the rate of its kind
    t_image, shape, bn_cts, sq_cts, first_conv = 0.0, (1, 28, 28), 0, 0, True
    for k_, _, a_ in topo:
        if k_ == "conv":
            t_image += layer_macs(k_, a_) / (rate["conv1"] if first_conv else rate.get("conv", rate["conv1"])); first_conv = False
        elif k_ == "fc":
            t_image += layer_macs(k_, a_) / rate.get("fc", rate.get("conv", rate["conv1"]))
        elif k_ in ("pool", "avgpool"):
            t_image += a_["zd"] * ((a_["xd"] - a_["xf"]) // a_["xs"] + 1) * ((a_["yd"] - a_["yf"]) // a_["ys"] + 1) / pooled_per_s
        elif k_ == "bn":
            bn_cts += int(np.prod(shape))
        elif k_ == "square":
            sq_cts += int(np.prod(shape))
        shape = out_shape(k_, a_, shape)
    if t_bn is not None:
        t_image += bn_cts * t_bn + sq_cts * t_sq / cores        # (SquareLayer splits its ciphertexts over th_count threads; BatchNormLayer and the pools are single-threaded)
        notes.append(f"one ciphertext through BatchNormLayer::forward in {t_bn * 1e3:.1f} ms (x {bn_cts} per image, single-threaded in the reference) and through SquareLayer::forward "
                     f"(Evaluator::square + relinearize, dbc 16) in {t_sq * 1e3:.1f} ms (x {sq_cts} per image / th_count={cores})")
    elif any(k_ in ("square", "bn") for k_, _, _ in topo):
        notes.append("square / batch-norm layers not sampled")
    total_macs = sum(layer_macs(k_, a_) for k_, _, a_ in topo)
    return dict(value=1.0 / t_image, unit="encrypted images/sec", cores=cores, kind="reference",
                sample="the compiled reference (SEAL 2.3.1 + CrCNN's layer classes, oracle/_ref/ref_harness) on pieces of image 0, one per layer kind, each layer of the network "
                       f"extrapolated by the multiply-accumulate rate of its kind ({total_macs} MACs/image): " + "; ".join(notes),
                mac_per_s=total_macs / t_image, mac_per_s_by_kind={k_: round(v, 1) for k_, v in rate.items()}, measured_in_full=False)


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
