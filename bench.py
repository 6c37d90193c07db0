#!/usr/bin/env python3
"""bench.py -- encrypted images/sec of the CrCNN evaluation path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

One "step" = one pass of the hot path (Network::forward: conv/pool/[bn/square]/dense over every image) over one batch of
`--batch` synthetic encrypted MNIST-like images per GPU.  Inputs are encrypted before the timed region and are resident in
HBM; the encoded (NTT-form) weights are built on rank 0 and broadcast with RCCL; images are sharded across ranks (weak
scaling, no data-path collective).  Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      achieved HBM GB/s of the dominant kernel (algorithmic bytes / measured duration) vs the 8 TB/s peak
  cpu_baseline  the CPU oracle executing the reference's own operation order on the host cores, on a bounded sample
  ms_per_layer  per-image milliseconds per layer (the reference's T_LAYER_i columns, mainparams.cpp:81)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CONFIGS = {
    # BASELINE.json configs[1]: PlainModelTiny.h5, n=4096, batch=1024 on one MI355X  (q = coeff_modulus_128(4096), t = 2^20)
    "tiny4096": dict(model="PlainModelTiny", n=4096, k=2, t=1 << 32, batch=1024, chunk=24),   # t=2^32: exact logits without the client-side refresh
    # configs[2]: ApproxPlainModel.h5, n=8192, 3 coeff moduli, batch=1024
    "approx8192": dict(model="ApproxPlainModel", n=8192, k=3, t=1 << 42, batch=1024, chunk=24),   # t=2^42: exact logits, 19 bits of budget left
    # configs[4]: PlainModelWoPad.h5, n=16384, 4 coeff moduli
    "wopad16384": dict(model="PlainModelWoPad", n=16384, k=4, t=1 << 44, batch=1024, chunk=6),
    # the coefficient modulus CrCNN itself would run at n=8192 (all four primes of coeff_modulus_128(8192)); at n=16384 the
    # eight default primes would need 424 GB for PlainModelWoPad's encoded weights alone (> HBM), so that one stays at k=4
    "approx8192k4": dict(model="ApproxPlainModel", n=8192, k=4, t=1 << 42, batch=1024, chunk=16),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--config", default="tiny4096", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="encrypted images per GPU per step (default: the config's)")
    ap.add_argument("--chunk", type=int, default=None, help="images processed per layer launch")
    ap.add_argument("--distinct", type=int, default=4, help="distinct encrypted images (tiled to the batch on device)")
    ap.add_argument("--mode", default="resident", choices=["resident", "layerwise"])
    ap.add_argument("--no-fuse", action="store_true", help="do not fold pooling layers into the preceding convolution")
    ap.add_argument("--unfused-images", type=int, default=128, help="images of the extra, untimed-for-`value` pass with every reference layer run separately")
    ap.add_argument("--t-bits", type=int, default=None, help="override the plain modulus t = 2^bits")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target size of the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def plain_forward(model, W, img):
    """float64 numpy forward of the plaintext network (PlainModel/*.py semantics) for the prediction check"""
    from crcnn_amd.netrun import TOPOLOGIES
    x = img.astype(np.float64)[None]
    for kind, name, a in TOPOLOGIES[model]:
        if kind == "conv":
            w = W[name + ".weight"].astype(np.float64).reshape(a["nf"], a["zd"], a["xf"], a["yf"]); b = W[name + ".bias"].astype(np.float64)
            xo, yo = (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1
            y = np.zeros((a["nf"], xo, yo))
            for i in range(xo):
                for j in range(yo):
                    patch = x[:, i * a["xs"]:i * a["xs"] + a["xf"], j * a["ys"]:j * a["ys"] + a["yf"]]
                    y[:, i, j] = (w * patch[None]).sum(axis=(1, 2, 3)) + b
            x = y
        elif kind in ("pool", "avgpool"):
            xo, yo = (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1
            y = np.zeros((a["zd"], xo, yo))
            for i in range(xo):
                for j in range(yo):
                    y[:, i, j] = x[:, i * a["xs"]:i * a["xs"] + a["xf"], j * a["ys"]:j * a["ys"] + a["yf"]].sum(axis=(1, 2))
            x = y / (a["xf"] * a["yf"]) if kind == "avgpool" else y
        elif kind == "bn":
            mean = W[name + ".running_mean"].astype(np.float64); var = W[name + ".running_var"].astype(np.float64)
            x = (x - mean[:, None, None]) / np.sqrt(var + 1e-5)[:, None, None]
        elif kind == "square":
            x = x * x
        elif kind == "fc":
            w = W[name + ".weight"].astype(np.float64).reshape(a["out_dim"], a["in_dim"]); b = W[name + ".bias"].astype(np.float64)
            x = (w @ x.reshape(-1) + b).reshape(1, a["out_dim"], 1)
    return x.reshape(-1)


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


def cpu_baseline_reference(cfg, q, W, x0, cores):
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
    t_conv, t_pool = float(rows[0][4].rstrip("us")) * 1e-6, float(rows[1][4].rstrip("us")) * 1e-6
    macs0 = layer_macs(k0, a0)
    total_macs = sum(layer_macs(k_, a_) for k_, _, a_ in topo)
    pool_cts = sum(a_["zd"] * ((a_["xd"] - a_["xf"]) // a_["xs"] + 1) * ((a_["yd"] - a_["yf"]) // a_["ys"] + 1) for k_, _, a_ in topo if k_ in ("pool", "avgpool"))
    cts1 = a1["zd"] * ((a1["xd"] - a1["xf"]) // a1["xs"] + 1) * ((a1["yd"] - a1["yf"]) // a1["ys"] + 1)
    t_image = total_macs / (macs0 / t_conv) + pool_cts * (t_pool / cts1)
    return dict(value=1.0 / t_image, unit="encrypted images/sec", cores=cores, kind="reference",
                sample=f"the compiled reference (SEAL 2.3.1 + CrCNN ConvolutionalLayer/{'Avg' if k1 == 'avgpool' else ''}PoolingLayer::forward, oracle/_ref/ref_harness) on image 0: "
                       f"{n0} {macs0} ct*pt MACs in {t_conv:.2f}s with th_count={cores}, {n1} in {t_pool:.2f}s (single-threaded in the reference); "
                       f"per-image time extrapolated by MAC count ({total_macs} MACs/image) and pooled-ciphertext count; square/bn layers not sampled",
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


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network, TOPOLOGIES, layer_macs

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)                     # (rehearsals with more ranks than GPUs share a device; the driver uses one rank per GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CRC_DIST_BACKEND", "nccl")           # "nccl" is RCCL on ROCm; "gloo" only for single-GPU rehearsals of this code path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cfg = dict(CONFIGS[args.config])
    if args.t_bits:
        cfg["t"] = 1 << args.t_bits
    B = args.batch or cfg["batch"]; C = min(args.chunk or cfg["chunk"], B)
    q = ca.default_coeff_modulus_128(cfg["n"])[:cfg["k"]]
    E = ca.Engine(cfg["n"], q, cfg["t"], device=local)
    local_rank_env = int(os.environ.get("LOCAL_RANK", "0"))
    E.stream = torch.cuda.current_stream().cuda_stream or None
    keep = []

    def alloc(nbytes):
        t = torch.empty((int(nbytes) + 7) // 8, dtype=torch.int64, device=dev); keep.append(t); return t

    model = cfg["model"]
    h5 = os.path.join(ROOT, "tests", "golden", "models", model + ".h5")
    W = {nm: ca.h5_read(h5, nm) for nm in ca.h5_list(h5) if not nm.endswith("num_batches_tracked")}

    # ---- keys + encrypted inputs (client side, untimed): `distinct` synthetic images encrypted on the host, tiled on device
    t_setup = time.time()
    sk, pk = E.keygen(2024)
    needs_evk = any(k_ == "square" for k_, _, _ in TOPOLOGIES[model])
    d_evk = None
    if needs_evk:
        evk = E.gen_evk(2025, sk)
        d_evk = alloc(evk.nbytes); d_evk.copy_(torch.from_numpy(evk.view(np.int64)))
    from crcnn_amd.synth import normalize, synth_image
    D = max(1, min(args.distinct, B))
    imgs = [normalize(synth_image(rank * 100003 + i)) for i in range(D)]
    ctw = 2 * E.k * E.n
    src = torch.empty((D, 784 * ctw), dtype=torch.int64, device=dev)
    for i, im in enumerate(imgs):
        pl, _ = E.encode(im.reshape(-1))
        ct = E.encrypt(pk, pl, 7000 + 1000 * i)
        src[i].copy_(torch.from_numpy(ct.reshape(-1).view(np.int64)))
    # the batch is `D` distinct encrypted images tiled B/D times.  It is materialised in HBM when it fits beside the weights
    # (Tiny: 98 GiB); for the bigger rings (1024 x 784 cts is 294 GiB at n=8192) a window of whole chunks is kept instead and
    # chunk c reads window position c mod window -- the same tiling, the same bytes per image
    img_bytes = 784 * ctw * 8
    free_b, total_b = torch.cuda.mem_get_info(dev)
    from crcnn_amd.netrun import TOPOLOGIES as _T
    est_weights = sum((a_.get("nf", 0) * a_.get("zd", 0) * a_.get("xf", 0) * a_.get("yf", 0) + a_.get("in_dim", 0) * a_.get("out_dim", 0)) for _, _, a_ in _T[model]) * E.k * E.n * 8
    budget = max(img_bytes * C, int(0.45 * (free_b - 1.25 * est_weights)))
    step_w = C * D // np.gcd(C, D)                      # window must be a multiple of the chunk and of the tiling period
    window = min(B, max(step_w, (budget // img_bytes) // step_w * step_w)) if budget // img_bytes < B else B
    x_all = alloc(window * img_bytes).view(window, 784 * ctw)
    for b0 in range(0, window, D):
        nb = min(D, window - b0); x_all[b0:b0 + nb].copy_(src[:nb])
    del src

    # ---- encoded weights: rank 0 encodes + NTTs, RCCL broadcast to the others
    net = Network(E, model, weights=W, alloc=alloc, resident=(args.mode == "resident"), d_evk=d_evk, materialize=(rank == 0), fuse_pool=False)
    torch.cuda.synchronize()
    bcast_s = 0.0
    if world > 1:
        from crcnn_amd import shard
        dist.barrier(); t0 = time.time()
        shard.broadcast_buffers([buf for buf, _ in net.param_bufs], src=0, chunk_bytes=1 << 30)      # RCCL over xGMI
        torch.cuda.synchronize(); bcast_s = time.time() - t0
    net.materialize = True                       # every rank now holds the encoded parameters (needed by fuse())
    out_all = alloc(B * 10 * ctw * 8).view(B, 10 * ctw)
    # ---- reference layer structure first (every CrCNN layer run as its own kernel sequence, NTT-resident): a short pass
    unfused = None
    want_fuse = args.mode == "resident" and not args.no_fuse
    if want_fuse and args.unfused_images > 0:
        nu = min(B, max(C, args.unfused_images // C * C))
        net.prepare(C)
        net.forward(x_all[0], 1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c0 in range(0, nu, C):
            d_out = net.forward(x_all[c0 % window], min(C, nu - c0))
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), min(C, nu - c0) * 10 * ctw * 8, E.stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        unfused = dict(images=nu, images_per_s=round(nu / dt, 3), ms_per_image=round(dt / nu * 1e3, 3), first_outputs=out_all[:min(D, nu)].clone(),
                       layers=[pl[1] for pl in net.plan])
        for t_ in list(net.buf) + [net.work]:          # give the large unfused activation buffers back before the main pass
            keep[:] = [k_ for k_ in keep if k_ is not t_]
        del net.buf, net.work, t_
        torch.cuda.empty_cache()
    if want_fuse:
        net.fuse()            # fold avg/sum pooling into the preceding convolution where that removes MACs (exact)
    net.prepare(C)
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    nl = len(net.plan)
    lay_ev = []

    def step(record):
        for c0 in range(0, B, C):
            cb = min(C, B - c0)
            evs = []

            def timer(i, name, kind, phase):
                if record:
                    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
            d_out = net.forward(x_all[c0 % window], cb, timer=timer)
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), cb * 10 * ctw * 8, E.stream)
            if record:
                lay_ev.append((cb, evs))

    # untimed module-load pass on a single image (not a step)
    net.forward(x_all[0], 1)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s == args.steps - 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        from crcnn_amd import shard
        elapsed = shard.max_over_ranks(elapsed, dev)

    # ---- per-layer times of the last step (HIP events on the launch stream)
    lay_ms = np.zeros(nl); lay_launch = np.zeros(nl); lay_cnt = np.zeros(nl)
    for cb, evs in lay_ev:
        for i in range(nl):
            ms = evs[2 * i].elapsed_time(evs[2 * i + 1])
            lay_ms[i] += ms
            if cb == C:
                lay_launch[i] += ms; lay_cnt[i] += 1
    ms_per_layer = {net.plan[i][1]: round(float(lay_ms[i] / B), 4) for i in range(nl)}

    # ---- verification outside the timed region: tiled images give identical outputs; decrypted logits match the plain model
    ok_tile = all(bool(torch.equal(out_all[b], out_all[b % D])) for b in range(D, B, max(1, (B - D) // 16)))
    if unfused is not None:     # folding pooling into the convolution must not change a single output bit
        fo = unfused.pop("first_outputs")
        unfused["outputs_identical_to_fused"] = bool(torch.equal(fo, out_all[:fo.shape[0]]))
    outs = out_all[:D].cpu().numpy().view(np.uint64).reshape(D, 10, 2, E.k, E.n)
    preds_ok, budgets, max_err = 0, [], 0.0
    for i in range(D):
        dec = E.decrypt(sk, outs[i])
        logits = np.array([E.decode(dec[j]) for j in range(10)])
        want = plain_forward(model, W, imgs[i])
        budgets.append(E.noise_budget(sk, outs[i][0]))
        max_err = max(max_err, float(np.abs(logits - want).max()))
        preds_ok += int(np.argmax(logits) == np.argmax(want))

    if world > 1:           # every rank must have verified its own outputs
        from crcnn_amd import shard
        all_ok = shard.gather_counts(int(ok_tile and preds_ok == D and (unfused is None or unfused["outputs_identical_to_fused"])), dev)
    else:
        all_ok = int(ok_tile and preds_ok == D)
    if rank != 0:
        dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: algorithmic bytes per launch / measured duration
    dom = int(np.argmax(lay_launch))
    kind, name, a, p, ishape, oshape = net.plan[dom]
    in_cts, out_cts = int(np.prod(ishape)), int(np.prod(oshape))
    ct_bytes = 8 * E.k * E.n * 2
    wbytes = layer_macs(kind, a) // max(1, out_cts // (a.get("nf", a.get("out_dim", 1)))) * 8 * E.k * E.n if kind in ("conv", "fc") else 0
    if kind == "conv":
        wbytes = a["nf"] * a["zd"] * a["xf"] * a["yf"] * 8 * E.k * E.n
    elif kind == "fc":
        wbytes = a["in_dim"] * a["out_dim"] * 8 * E.k * E.n
    alg_bytes = C * (in_cts + out_cts) * ct_bytes + wbytes
    dur_ms = lay_launch[dom] / max(1, lay_cnt[dom])
    achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
    macs_launch = layer_macs(kind, a) * C
    # HBM traffic of that launch from rocprofv3 PMC passes (FETCH_SIZE corrected x2 for gfx950, WRITE_SIZE), collected offline with
    # tools/bench_mac.py and committed under profiles/ -- bench.py cannot run the profiler on itself
    traffic = None
    kernel_label = f"mac3_kernel ({name}, {C} images/launch)" if kind in ("conv", "fc") else f"{kind} ({name})"
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"))).get(args.config)
        if pm and pm["kernel"] == kernel_label:
            traffic = int(pm["traffic_bytes"])
    except Exception:
        pass
    roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                    kernel=kernel_label if kind in ("conv", "fc") else f"{kind} ({name})",
                    launch_ms=round(float(dur_ms), 3), algorithmic_bytes_per_launch=int(alg_bytes),
                    modmul_per_s=round(macs_launch * 2 * E.k * E.n / (dur_ms * 1e-3), 1) if dur_ms > 0 else None)

    cpu = None
    if args.cpu_seconds > 0:
        x0 = x_all[0].cpu().numpy().view(np.uint64).reshape(1, 28, 28, 2, E.k, E.n)
        cpu = cpu_baseline_reference(cfg, q, W, x0, host_cores()) or cpu_baseline(cfg, q, W, x0, args.cpu_seconds)
        cpu["value"] = round(cpu["value"], 6); cpu["mac_per_s"] = round(cpu["mac_per_s"], 1)

    total_images = B * world * args.steps
    value = total_images / elapsed
    line = {
        "metric": "encrypted images/sec", "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": f"synthetic ({D} distinct MNIST-like encrypted images per GPU tiled to the batch" + ("" if window == B else f", resident as a {window}-image window") + f"; trained weights from {model}.h5)",
        "config": {"workload": f"{model}.h5 n={cfg['n']} k={cfg['k']} t=2^{cfg['t'].bit_length() - 1} batch={B}/GPU chunk={C} ({args.config}, BASELINE configs)",
                   "mode": args.mode + ("+conv/pool folding" if want_fuse else ""), "parallelism": f"image-sharded x{world}, RCCL weight broadcast"},
        "ms_per_layer": ms_per_layer, "reference_layer_structure": unfused, "roofline": roofline, "cpu_baseline": cpu,
        "check": {"tiled_outputs_identical": bool(ok_tile), "predictions_match_plain_model": f"{preds_ok}/{D}", "max_logit_abs_err": round(max_err, 6),
                  "noise_budget_bits": budgets, "ranks_verified": f"{all_ok}/{world}"},
        "setup_s": round(setup_s, 1), "weight_broadcast_s": round(bcast_s, 2), "weight_bytes": int(net.weight_bytes),
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
