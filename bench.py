#!/usr/bin/env python3
"""bench.py -- encrypted images/sec of the CrCNN evaluation path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

One "step" = one pass of the hot path (Network::forward: conv/pool/[bn/square]/dense over every image) over one batch of
`--batch` synthetic encrypted MNIST-like images per GPU.  Inputs are encrypted before the timed region and are resident in
HBM; the encoded (NTT-form) weights are built on rank 0 and broadcast with RCCL; images are sharded across ranks (weak
scaling, no data-path collective).  Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel against the roof that bounds it: int8 matrix-core TOP/s for the limb GEMM (frac = executed, useful_frac = without channel /
                filter padding; the HBM view of the same launch beside it), HBM GB/s (algorithmic bytes / measured duration vs the 8 TB/s peak) for everything else
  cpu_baseline  the compiled reference (oracle/_ref/ref_harness; the CPU oracle where that binary is absent) on the host cores, on a bounded sample: conv1 + pool1 of
                image 0, one batch-norm ciphertext and one Square ciphertext, extrapolated by MAC / ciphertext counts
  ms_per_layer  per-image milliseconds per (fused) layer; reference_layer_structure.production_kernels.T_LAYER = the reference's own T_LAYER_i columns
                (mainparams.cpp:81: one per layer of the unfused network) on the same kernels
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INT8_PEAK_TOPS = 5000.0        # dense int8 MFMA peak: 2x the ~2.5 PF bf16 rate per clock (MI355X_MICROARCH.md, matrix cores; no sparsity)

CONFIGS = {
    # BASELINE.json configs[1]: PlainModelTiny.h5, n=4096, batch=1024 on one MI355X  (q = coeff_modulus_128(4096), t = 2^20)
    "tiny4096": dict(model="PlainModelTiny", n=4096, k=2, t=1 << 32, batch=1024, chunk=128),   # t=2^32: exact logits without the client-side refresh
    # configs[2]: ApproxPlainModel.h5, n=8192, 3 coeff moduli, batch=1024
    "approx8192": dict(model="ApproxPlainModel", n=8192, k=3, t=1 << 42, batch=1024, chunk=32, tail=2),   # t=2^42: exact logits, 19 bits of budget left; dense layers per 64 images (+1 %)
    # configs[4]: PlainModelWoPad.h5, n=16384, 4 coeff moduli
    # (tail=4: the dense layers run once per 4 chunks = 24 images -- two-level chunking, netrun.prepare: fc3 streams 177 GiB of limb-form weights per launch)
    "wopad16384": dict(model="PlainModelWoPad", n=16384, k=4, t=1 << 44, batch=1024, chunk=6, tail=4),
    # the coefficient modulus CrCNN itself would run at n=8192 (all four primes of coeff_modulus_128(8192)); at n=16384 the
    # eight default primes would need 424 GB for PlainModelWoPad's encoded weights alone (> HBM), so that one stays at k=4
    "approx8192k4": dict(model="ApproxPlainModel", n=8192, k=4, t=1 << 42, batch=1024, chunk=16, tail=2),      # (dense layers per 32 images: a full 64-row tile, +7 %)
    # every prime of coeff_modulus_128(16384), the coefficient modulus CrCNN's own setParameters(16384, t) picks: 424 GB of NTT-form weights -- fc3 keeps
    # coefficient-form plaintexts in HBM and is lifted + transformed a filter tile at a time inside every forward (netrun: streamed layers)
    "wopad16384k8": dict(model="PlainModelWoPad", n=16384, k=8, t=1 << 44, batch=96, chunk=4, tail=8),
    # small ring for the tests of this script and single-GPU rehearsals of the multi-rank path (golden: net_tiny1024_eng.json)
    "tiny1024": dict(model="PlainModelTiny", n=1024, k=2, q=[0x7fffffff380001, 0x3fffffff000001], t=1 << 32, batch=48, chunk=24),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--config", default="tiny4096", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="encrypted images per GPU per step (default: the config's)")
    ap.add_argument("--chunk", type=int, default=None, help="images processed per layer launch")
    ap.add_argument("--tail", type=int, default=None, help="chunks per launch of the dense layers (two-level chunking; default: the config's)")
    ap.add_argument("--distinct", type=int, default=4, help="distinct encrypted images (tiled to the batch on device)")
    ap.add_argument("--mode", default="resident", choices=["resident", "layerwise"])
    ap.add_argument("--no-fuse", action="store_true", help="do not fold pooling layers into the preceding convolution")
    ap.add_argument("--unfused-images", type=int, default=128, help="images of the extra, untimed-for-`value` pass with every reference layer run separately")
    ap.add_argument("--t-bits", type=int, default=None, help="override the plain modulus t = 2^bits")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--also", default="auto", help="a second workload measured in the same invocation and reported under \"also\" (auto: approx8192 = BASELINE "
                    "configs[2]/[3] beside the tiny4096 headline; none: skip)")
    ap.add_argument("--also-steps", type=int, default=2)
    ap.add_argument("--host-cpp", type=int, default=1, help="1: also time the same workload through the C++ host classes (crcnn_amd/lib/bench_host) and check it against the Python twin (models below 60 GiB of encoded weights; 2: any model)")
    ap.add_argument("--host-cpp-steps", type=int, default=3)
    ap.add_argument("--also-batch", type=int, default=None)
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, rendezvous (gloo, no GPU call) and report: CPU test of the self-launch path")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (torch.distributed.run, one per GPU) as a CHILD process -- this
    parent has made no GPU call and never execs -- relay rank 0's JSON line, and exit non-zero if any rank does."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC (RCCL across processes)
    env["CRC_SELF_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n"); rc = 1
    return rc


def launch_check(args):
    """ranks rendezvous over gloo and count themselves; nothing touches a GPU (tests/test_multiproc_gloo.py)"""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    seen = world
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([1], dtype=torch.int64); dist.all_reduce(t); seen = int(t.item())
        dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": seen, "self_launched": bool(os.environ.get("CRC_SELF_LAUNCHED"))}), flush=True)
    return 0 if seen == args.gpus else 1


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


# golden fixtures (tests/golden/net_*.json, produced by the compiled reference: oracle/make_golden_nets.py) whose encrypted input is what
# this script generates for image 0 of rank 0 -- same parameter set, same seeded client side
GOLDEN_FOR = {"tiny4096": "tiny4096_t32", "approx8192": "approx8192_t42", "approx8192k4": "approx8192k4_t42", "wopad16384": "wopad16384_t44", "wopad16384k8": "wopad16384k8_t44", "tiny1024": "tiny1024_eng"}
KEY_SEED, EVK_SEED, ENC_SEED = 2024, 2025, 7000


def golden_check(cfg_name, cfg, q, rank, x0_sha, out0_sha):
    """True / False when a reference-made golden exists for exactly these parameters and inputs, else None"""
    path = os.path.join(ROOT, "tests", "golden", f"net_{GOLDEN_FOR.get(cfg_name, '')}.json")
    if rank != 0 or not os.path.exists(path):
        return None, None
    g = json.load(open(path))
    same = (g.get("input_gen") == "engine" and g["model"] == cfg["model"] and g["n"] == cfg["n"] and g["t"] == cfg["t"] and [int(v) for v in g["q"]] == [int(v) for v in q]
            and (g["key_seed"], g["evk_seed"], g["enc_seed"], g["image_index"]) == (KEY_SEED, EVK_SEED, ENC_SEED, 0))
    if not same:
        return None, None
    return bool(g["input_sha256"] == x0_sha and g["out_sha256"] == out0_sha), os.path.basename(path)


class Dist:
    """rank bookkeeping + the two kinds of collective this script needs: the weight broadcast (RCCL through the engine's own C ABI,
    crc_comm_* / crc_broadcast_weights) and tiny host-side reductions (timing, verification counts)"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0")); self.world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        ndev = torch.cuda.device_count()
        self.backend = os.environ.get("CRC_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm; "gloo" only for single-GPU rehearsals of this code path
        if self.world > 1 and self.backend == "nccl" and ndev < self.world:
            raise SystemExit(f"--gpus {self.world} needs {self.world} GPUs, {ndev} visible (CRC_DIST_BACKEND=gloo rehearses the multi-rank path on fewer)")
        self.local = local % max(1, ndev)                # (rehearsals with more ranks than GPUs share a device; the driver uses one rank per GPU)
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.comm = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev, timeout=datetime.timedelta(minutes=10))      # a collective nobody else joins ends the job, not hangs it
            else:
                dist.init_process_group(self.backend, timeout=datetime.timedelta(minutes=10))

    def make_comm(self, E):
        """RCCL communicator of the engine (C ABI); the 128-byte rendezvous id travels over the torch.distributed store"""
        if self.world == 1 or self.backend != "nccl":
            return None, None
        comm, err = None, None
        # every rank runs the SAME sequence of torch.distributed collectives whatever fails: rank 0 always broadcasts (id or None, error), every rank skips
        # crc_comm_create when there is no id, and the all-reduce afterwards tells everybody whether ALL ranks hold a communicator (otherwise all of them use
        # the torch.distributed broadcast).  A rank that dies inside ncclCommInitRank is caught by the process group's timeout (init_process_group above).
        obj = [None, None]
        if self.rank == 0:
            try:
                obj = [E.comm_unique_id(), None]
            except Exception as ex:
                obj = [None, f"{type(ex).__name__}: {ex}"]
        self.dist.broadcast_object_list(obj, src=0)
        if obj[0] is None:
            err = obj[1] or "rank 0 could not make a rendezvous id"
        else:
            try:
                comm = E.comm_create(self.world, self.rank, obj[0])
            except Exception as ex:
                err = f"{type(ex).__name__}: {ex}"
        have = self.sum(int(comm is not None))
        if have != self.world:
            if comm is not None:
                E.comm_destroy(comm)
            errs = [e_ for e_ in self.gather(err) if e_]
            return None, (errs[0] if errs else "communicator missing on some rank")
        return comm, None

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, v):
        if self.world == 1:
            return v
        from crcnn_amd import shard
        return shard.max_over_ranks(v, self.dev)

    def sum(self, v):
        if self.world == 1:
            return v
        from crcnn_amd import shard
        return shard.gather_counts(v, self.dev)

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


def run_config(args, D_, cfg_name, steps, warmup, batch=None, chunk=None, full=True):
    """one workload: setup (keys, encrypted inputs, encoded weights + broadcast), `warmup` + `steps` timed passes, verification.
    full=False: the secondary workload (no reference-layer-structure pass, no CPU baseline).  Returns (result dict on rank 0, all checks ok)"""
    import torch
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network, TOPOLOGIES, layer_macs
    rank, world, dev = D_.rank, D_.world, D_.dev

    cfg = dict(CONFIGS[cfg_name])
    if args.t_bits and full:
        cfg["t"] = 1 << args.t_bits
    B = batch or cfg["batch"]; C = min(chunk or cfg["chunk"], B)
    G = max(1, min(int(args.tail or cfg.get("tail", 1)), B // C))       # chunks per run of the dense layers (two-level chunking)
    q = cfg.get("q") or ca.default_coeff_modulus_128(cfg["n"])[:cfg["k"]]
    E = ca.Engine(cfg["n"], q, cfg["t"], device=D_.local)
    E.stream = torch.cuda.current_stream().cuda_stream or None
    keep = []

    def alloc(nbytes):
        if os.environ.get("CRC_BENCH_ALLOC_LOG") and nbytes > (1 << 30):          # (debugging aid: where the HBM goes)
            import traceback
            fr = traceback.extract_stack(limit=3)[0]
            sys.stderr.write(f"alloc {nbytes / 2**30:8.2f} GiB  {os.path.basename(fr.filename)}:{fr.lineno} {fr.name}   (torch allocated {torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB)\n")
        t = torch.empty((int(nbytes) + 7) // 8, dtype=torch.int64, device=dev); keep.append(t); return t

    model = cfg["model"]
    h5 = os.path.join(ROOT, "tests", "golden", "models", model + ".h5")
    W = {nm: ca.h5_read(h5, nm) for nm in ca.h5_list(h5) if not nm.endswith("num_batches_tracked")}

    # ---- keys + encrypted inputs (client side, untimed): `distinct` synthetic images encrypted on the host, tiled on device.
    # Seeded (deterministic, NOT secure) on purpose: image 0 of rank 0 is the input of the reference-made golden (GOLDEN_FOR)
    t_setup = time.time()
    sk, pk = E.keygen(KEY_SEED)
    needs_evk = any(k_ == "square" for k_, _, _ in TOPOLOGIES[model])
    d_evk = evk = None
    if needs_evk:
        evk = E.gen_evk(EVK_SEED, sk)
        d_evk = alloc(evk.nbytes); d_evk.copy_(torch.from_numpy(evk.view(np.int64)))
    from crcnn_amd.synth import normalize, synth_image
    D = max(1, min(args.distinct, B))
    imgs = [normalize(synth_image(rank * 100003 + i)) for i in range(D)]
    ctw = 2 * E.k * E.n
    src = torch.empty((D, 784 * ctw), dtype=torch.int64, device=dev)
    x0_sha = None
    for i, im in enumerate(imgs):
        pl, _ = E.encode(im.reshape(-1))
        ct = E.encrypt(pk, pl, ENC_SEED + 1000 * i)
        if i == 0:
            import hashlib
            x0_sha = hashlib.sha256(np.ascontiguousarray(ct).tobytes()).hexdigest()
        src[i].copy_(torch.from_numpy(ct.reshape(-1).view(np.int64)))
    # the batch is `D` distinct encrypted images tiled B/D times.  It is materialised in HBM when it fits beside the weights
    # (Tiny: 98 GiB); for the bigger rings (1024 x 784 cts is 294 GiB at n=8192) a window of whole chunks is kept instead and
    # chunk c reads window position c mod window -- the same tiling, the same bytes per image
    img_bytes = 784 * ctw * 8
    free_b, total_b = torch.cuda.mem_get_info(dev)
    est_weights = sum((a_.get("nf", 0) * a_.get("zd", 0) * a_.get("xf", 0) * a_.get("yf", 0) + a_.get("in_dim", 0) * a_.get("out_dim", 0)) for _, _, a_ in TOPOLOGIES[model]) * E.k * E.n * 8
    budget = max(img_bytes * C, int(0.45 * (free_b - 2.3 * est_weights)))          # (2.3: the limb copy of the weights is built beside the canonical one)
    step_w = C * D // np.gcd(C, D)                      # window must be a multiple of the chunk and of the tiling period
    window = min(B, max(step_w, (budget // img_bytes) // step_w * step_w)) if budget // img_bytes < B else B
    x_all = alloc(window * img_bytes).view(window, 784 * ctw)
    for b0 in range(0, window, D):
        nb = min(D, window - b0); x_all[b0:b0 + nb].copy_(src[:nb])
    del src

    # ---- encoded weights: rank 0 encodes + NTTs, RCCL broadcast to the others, every rank checksums what it holds
    net = Network(E, model, weights=W, alloc=alloc, resident=(args.mode == "resident"), d_evk=d_evk, materialize=(rank == 0), fuse_pool=False)

    def release(buf):
        keep[:] = [k_ for k_ in keep if k_ is not buf]
    net.release = release
    torch.cuda.synchronize()
    bcast = None
    if world > 1:
        bufs = list(net.param_bufs) + ([(d_evk, d_evk.numel() * 8)] if d_evk is not None else [])
        nbytes = sum(n_ for _, n_ in bufs)
        comm, comm_err = D_.make_comm(E)
        D_.barrier(); t0 = time.time()
        if comm is not None:
            for buf, n_ in bufs:
                E.broadcast_weights(comm, buf, (n_ + 7) // 8 * 8, root=0)          # crc_broadcast_weights: ncclBroadcast over xGMI, <= 1 GiB pieces
            via = "crc_broadcast_weights (RCCL through the engine's C ABI)"
        else:
            from crcnn_amd import shard
            shard.broadcast_buffers([buf for buf, _ in bufs], src=0, chunk_bytes=1 << 30)
            via = f"torch.distributed ({D_.backend})" + (" -- rehearsal backend" if D_.backend != "nccl" else f" -- FALLBACK, crc_comm_create failed: {comm_err}")
        torch.cuda.synchronize(); D_.barrier(); bcast_s = time.time() - t0
        cs = [0, 0]
        for buf, n_ in bufs:
            x_, s_ = E.checksum64(buf, n_ // 8 * 8)
            cs[0] ^= x_; cs[1] = (cs[1] * 0x9E3779B97F4A7C15 + s_) & ((1 << 64) - 1)
        if comm is not None:
            allcs = [tuple(int(v) for v in row) for row in E.allgather_u64(comm, cs)]
            E.comm_destroy(comm)
        else:
            allcs = [tuple(c_) for c_ in D_.gather(cs)]
        bcast = dict(seconds=round(bcast_s, 3), bytes=int(nbytes), GBps=round(nbytes / bcast_s / 1e9, 2), via=via,
                     xgmi_link_peak_GBps=153.0, checksums_match=f"{sum(1 for c_ in allcs if c_ == allcs[0])}/{world}", checksum=f"{allcs[0][0]:016x}:{allcs[0][1]:016x}")
    net.materialize = True                       # every rank now holds the encoded parameters (needed by fuse())
    out_all = alloc(B * 10 * ctw * 8).view(B, 10 * ctw)
    # ---- reference layer structure first (every CrCNN layer run as its own kernel sequence, NTT-resident): a short pass
    unfused = None
    want_fuse = args.mode == "resident" and not args.no_fuse
    tilewise = any(pl[3].get("tilewise") for pl in net.plan)       # (their limb weights would have to be built twice, before and after the folding: skipped)
    prod_ref = None
    est_w_gib = est_weights / 2**30
    if want_fuse and full and args.unfused_images > 0 and not tilewise and world == 1 and est_w_gib < 60:      # (a SECOND encoded network is resident during this pass)
        # the reference's OWN layer structure (no folding: every CrCNN layer its own kernel sequence, NTT-resident) on the PRODUCTION kernels (matrix cores): the
        # reference's T_LAYER_0..N columns (mainparams.cpp:81) for this engine.  A second encoded network (the limb conversion drops the canonical weights that fuse()
        # needs), freed again before the main pass
        mark = len(keep)
        net_r = Network(E, model, weights=W, alloc=alloc, resident=True, d_evk=d_evk, materialize=True, fuse_pool=False)
        net_r.release = release
        Cu = min(C, 16)                                # (two encoded networks are resident during this pass: a smaller chunk of unfused activations)
        nu = min(B, max(Cu, args.unfused_images // Cu * Cu))
        net_r.prepare(Cu, limb=True)
        net_r.forward(x_all[0], 1); torch.cuda.synchronize()
        lev = []

        def timer_r(i, name, kind, phase):
            e = torch.cuda.Event(enable_timing=True); e.record(); lev.append((i, e))
        t0 = time.perf_counter()
        for c0 in range(0, nu, Cu):
            d_out = net_r.forward(x_all[c0 % window], min(Cu, nu - c0), timer=timer_r)
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), min(Cu, nu - c0) * 10 * ctw * 8, E.stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tl = np.zeros(len(net_r.plan))
        for j in range(0, len(lev), 2):
            tl[lev[j][0]] += lev[j][1].elapsed_time(lev[j + 1][1])
        prod_ref = dict(images=nu, images_per_s=round(nu / dt, 3), ms_per_image=round(dt / nu * 1e3, 3), first_outputs=out_all[:min(D, nu)].clone(),
                        T_LAYER={f"T_LAYER_{i}": round(float(tl[i] / nu), 4) for i in range(len(net_r.plan))},
                        layers=[pl[1] for pl in net_r.plan],
                        mac_kernel_per_layer={pl[1]: {ca.NTTL: "mfma_mac2w_kernel", ca.NTTL1: "mfma_conv1_kernel"}.get(pl[3].get("w_form"), "mac3_kernel") for pl in net_r.plan if pl[0] in ("conv", "fc")},
                        note="ms per image and layer of the UNFUSED network in the column order of the reference's timing rows (mainparams.cpp:81), matrix-core kernels")
        del net_r, lev, d_out
        del keep[mark:]
        import gc
        gc.collect(); torch.cuda.empty_cache()
    if want_fuse and full and args.unfused_images > 0 and not tilewise:
        Cu = min(C, 32)                                # the unfused conv1 output is 18 432 ciphertexts per image (Tiny): a smaller chunk than the main pass
        nu = min(B, max(Cu, args.unfused_images // Cu * Cu))
        net.prepare(Cu, limb=False)                    # (the limb conversion drops the canonical weights fuse() needs; it happens in the final prepare)
        net.forward(x_all[0], 1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c0 in range(0, nu, Cu):
            d_out = net.forward(x_all[c0 % window], min(Cu, nu - c0))
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), min(Cu, nu - c0) * 10 * ctw * 8, E.stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        unfused = dict(images=nu, images_per_s=round(nu / dt, 3), ms_per_image=round(dt / nu * 1e3, 3), first_outputs=out_all[:min(D, nu)].clone(),
                       layers=[pl[1] for pl in net.plan], kernels="vector ALU only (mac3_kernel: the canonical weights are still needed by the folding that follows)")
        for t_ in list(net.buf) + [net.work]:          # give the large unfused activation buffers back before the main pass
            keep[:] = [k_ for k_ in keep if k_ is not t_]
        del net.buf, net.work, t_
        torch.cuda.empty_cache()
    if want_fuse:
        net.fuse()            # fold avg/sum pooling into the preceding convolution where that removes MACs (exact)
    net.prepare(C, tail_group=G)
    G = net.G
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    nl = len(net.plan)
    lay_ev = []
    # groups of up to G full chunks (the dense layers run once per group); a ragged last chunk is a group of its own
    groups, c0 = [], 0
    while c0 < B:
        cb = min(C, B - c0)
        ng = min(G, (B - c0) // C) if cb == C else 1
        groups.append((c0, cb, max(1, ng))); c0 += cb * max(1, ng)

    def step(record):
        for (c0, cb, ng) in groups:
            evs = []

            def timer(i, name, kind, phase):
                if record:
                    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((i, e))
            d_out = net.forward_group([x_all[(c0 + j * cb) % window] for j in range(ng)], cb, timer=timer)
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), ng * cb * 10 * ctw * 8, E.stream)
            if record:
                lay_ev.append((cb, ng, evs))

    # untimed module-load pass on a single image (not a step)
    net.forward_group([x_all[0]], 1)
    torch.cuda.synchronize()
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    D_.barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        step(s == steps - 1)
    torch.cuda.synchronize()
    D_.barrier()
    elapsed = D_.max(time.perf_counter() - t0)

    # ---- per-layer times of the last step (HIP events on the launch stream)
    lay_ms = np.zeros(nl); lay_launch = np.zeros(nl); lay_cnt = np.zeros(nl)
    for cb, ng, evs in lay_ev:
        for j in range(0, len(evs), 2):                  # (layer i, start), (layer i, end) -- a head layer once per chunk of the group, a dense layer once per group
            i = evs[j][0]; ms = evs[j][1].elapsed_time(evs[j + 1][1])
            lay_ms[i] += ms
            if cb == C and (i < net.split or ng == G):
                lay_launch[i] += ms; lay_cnt[i] += 1
    ms_per_layer = {net.plan[i][1]: round(float(lay_ms[i] / B), 4) for i in range(nl)}

    # ---- verification outside the timed region: tiled images give identical outputs; decrypted logits match the plain model;
    # the output ciphertexts of image 0 are, bit for bit, the compiled reference's (golden fixture)
    ok_tile = all(bool(torch.equal(out_all[b], out_all[b % D])) for b in range(D, B, max(1, (B - D) // 16)))
    ok_fused = True
    if unfused is not None:     # folding pooling into the convolution must not change a single output bit
        fo = unfused.pop("first_outputs")
        ok_fused = unfused["outputs_identical_to_fused"] = bool(torch.equal(fo, out_all[:fo.shape[0]]))
    if prod_ref is not None:
        fo = prod_ref.pop("first_outputs")
        prod_ref["outputs_identical_to_fused"] = bool(torch.equal(fo, out_all[:fo.shape[0]]))
        ok_fused = ok_fused and prod_ref["outputs_identical_to_fused"]
        if unfused is None:
            unfused = {}
        unfused["production_kernels"] = prod_ref
    outs = out_all[:D].cpu().numpy().view(np.uint64).reshape(D, 10, 2, E.k, E.n)
    import hashlib
    gold_ok, gold_name = golden_check(cfg_name, cfg, q, rank, x0_sha, hashlib.sha256(np.ascontiguousarray(outs[0]).tobytes()).hexdigest())
    # BASELINE configs[0] in full (tests/golden/c1_tiny4096_t32.json: 32 images through the compiled reference): this run's distinct images ARE its first images
    c1_ok = None
    c1_path = os.path.join(ROOT, "tests", "golden", "c1_tiny4096_t32.json")
    if rank == 0 and cfg_name == "tiny4096" and os.path.exists(c1_path):
        c1 = json.load(open(c1_path))
        if (c1["t"], [int(v) for v in c1["q"]], c1["key_seed"], c1["enc_seed_base"], c1["enc_seed_stride"]) == (cfg["t"], [int(v) for v in q], KEY_SEED, ENC_SEED, 1000):
            have = [i for i in range(D) if str(i) in c1["images"]]
            hits = sum(1 for i in have if c1["images"][str(i)]["out_sha256"] == hashlib.sha256(np.ascontiguousarray(outs[i]).tobytes()).hexdigest())
            c1_ok = f"{hits}/{len(have)}"
    preds_ok, budgets, max_err = 0, [], 0.0
    for i in range(D):
        dec = E.decrypt(sk, outs[i])
        logits = np.array([E.decode(dec[j]) for j in range(10)])
        want = plain_forward(model, W, imgs[i])
        budgets.append(E.noise_budget(sk, outs[i][0]))
        max_err = max(max_err, float(np.abs(logits - want).max()))
        preds_ok += int(np.argmax(logits) == np.argmax(want))
    mine_ok = bool(ok_tile and ok_fused and preds_ok == D and gold_ok is not False and (c1_ok is None or c1_ok.split("/")[0] == c1_ok.split("/")[1]))
    ranks_ok = D_.sum(int(mine_ok))               # every rank must have verified its own outputs
    all_ok = ranks_ok == world and (bcast is None or bcast["checksums_match"] == f"{world}/{world}")

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: algorithmic bytes per launch / measured duration
        dom = int(np.argmax(lay_launch))
        kind, name, a, p, ishape, oshape = net.plan[dom]
        in_cts, out_cts = int(np.prod(ishape)), int(np.prod(oshape))
        ct_bytes = 8 * E.k * E.n * 2
        wbytes = 0
        if kind == "conv":
            wbytes = a["nf"] * a["zd"] * a["xf"] * a["yf"] * 8 * E.k * E.n
        elif kind == "fc":
            wbytes = a["in_dim"] * a["out_dim"] * 8 * E.k * E.n
        CL = C * G if dom >= net.split else C           # images per launch of that layer
        alg_bytes = CL * (in_cts + out_cts) * ct_bytes + wbytes
        dur_ms = lay_launch[dom] / max(1, lay_cnt[dom])
        layer_ms = dur_ms
        kernel_note = None
        if p.get("w_form") == ca.NTTL and kind in ("conv", "fc") and dom > 0 and p["in_form"] != ca.NTTL:
            # the layer call is three kernels (tensor -> limb form, mfma_mac_kernel, result -> the next layer's form).  Time the MAC kernel itself: regenerate the
            # layer's input for one chunk with the layers in front of it, convert it once, then launch the layer on the limb tensor (HIP events, same stream)
            try:
                cur = x_all[0]
                for li in range(dom):
                    k_, n_, a_, p_, is_, os_ = net.plan[li]
                    assert k_ == "conv"
                    E.conv2d(cur, p_["w"], p_["b"], C, a_["zd"], a_["xd"], a_["yd"], a_["xs"], a_["ys"], a_["xf"], a_["yf"], a_["nf"], p_["in_form"], p_["out_form"], net.buf[net.slots[li]],
                             net.work, w_form=p_.get("w_form", ca.NTT))
                    cur = net.buf[net.slots[li]]
                gz = (a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"]) if kind == "conv" else (a["in_dim"], 1, 1, 1, 1, 1, 1, a["out_dim"])
                xl = alloc(E.limb_tensor_bytes(C, gz[0], gz[1], gz[2]))
                E.limb_pack_tensor(cur, p["in_form"], C, gz[0], gz[1], gz[2], xl)
                outk = net.buf[net.slots[dom]]
                run_k = lambda: E.conv2d(xl, p["w"], p["b"], C, gz[0], gz[1], gz[2], gz[3], gz[4], gz[5], gz[6], gz[7], ca.NTTL, ca.NTTL, outk, net.work, w_form=ca.NTTL)
                run_k(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run_k()
                e1.record(); torch.cuda.synchronize()
                dur_ms = e0.elapsed_time(e1) / 5
                kernel_note = (f"mfma_mac2w_kernel timed on its own (+ the 3 % slotmajor_to_limb conversion behind it): 5 launches on the limb-form input of one chunk; the whole layer call "
                               f"(limb_pack_tensor + mfma_mac2w_kernel + conversion) takes {layer_ms:.2f} ms inside the timed region")
            except Exception as ex:          # keep the layer-level figure
                kernel_note = f"kernel-only timing failed ({type(ex).__name__}); launch_ms is the whole layer call"
        if kernel_note is None and p.get("w_form") == ca.NTTL:
            kernel_note = ("HIP events around the layer call inside the timed region; the input arrives in limb form from the layer in front and the call is " +
                           ("mfma_mac2w_kernel alone (it writes the next dense layer's limb tensor itself)" if p["out_form"] == ca.NTTL else "mfma_mac2w_kernel + the conversion of its slot-major result (3-9 % of the call)"))
        achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        macs_launch = layer_macs(kind, a) * CL
        # HBM traffic of that launch: rocprofv3 PMC passes (FETCH_SIZE corrected x2 for gfx950, WRITE_SIZE) collected OFFLINE with
        # tools/bench_mac.py and committed under profiles/ -- bench.py cannot run the profiler on itself, so this is not measured in this run
        traffic, traffic_source = None, None
        kname = "mfma_mac2w_kernel" if p.get("w_form") == ca.NTTL else "mfma_conv1_kernel" if p.get("w_form") == ca.NTTL1 else "mac3_kernel"
        kernel_label = f"{kname} ({name}, {CL} images/launch)" if kind in ("conv", "fc") else f"{kind} ({name})"
        for pf in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", pf))).get(cfg_name)
                if pm and pm.get("per_ciphertext") and kind == "square":      # the Square + relinearise sequence: PMC bytes per ciphertext x the launch's ciphertexts
                    traffic = int(pm["traffic_bytes_per_ciphertext"] * CL * in_cts)
                    traffic_source = f"profiles/{pf} ({pm['kernel']}), offline rocprofv3 --pmc passes of tools/bench_square.py on the same ring (not measured in this run)"
                    break
                if pm and pm["kernel"] == kernel_label:
                    traffic = int(pm["traffic_bytes"]); traffic_source = f"profiles/{pf}, offline rocprofv3 --pmc passes of the same launch (not measured in this run)"
                    break
            except Exception:
                pass
        modmul_s = macs_launch * 2 * E.k * E.n / (dur_ms * 1e-3) if dur_ms > 0 and macs_launch else None
        if kname == "mfma_mac2w_kernel" and modmul_s:
            # the matrix-core kernel is bound by the int8 MFMA rate, not by HBM: 49 limb products (98 int8 operations) per modular multiply-add, against the dense
            # int8 peak (2x the bf16 rate per clock: MI355X_MICROARCH.md, matrix cores).  The HBM view of the same launch stays beside it
            tops = modmul_s * 98 / 1e12
            # executed = what the matrix cores really multiply (limb_exec_over_useful); useful = the layer's own multiply-adds
            exec_over_useful = limb_exec_over_useful(kind, a, CL, out_cts)
            roofline = dict(bound="mfma", achieved=round(tops * exec_over_useful, 1), peak=INT8_PEAK_TOPS, unit="TOP/s (int8)", frac=round(tops * exec_over_useful / INT8_PEAK_TOPS, 5),
                            useful_achieved=round(tops, 1), useful_frac=round(tops / INT8_PEAK_TOPS, 5), traffic=traffic,
                            ops="int8 multiply and add, 98 per modular multiply-add (7 x 7 balanced base-256 limb products); useful = the layer's ct x pt multiply-adds x 2 polys x k n, "
                                "executed = the same with rows / channels / filters padded to the kernel's tiles",
                            hbm_achieved_GBps=round(achieved, 2), hbm_frac=round(achieved / HBM_PEAK_GBS, 5))
        else:
            roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic)
        roofline.update(traffic_source=traffic_source, kernel=kernel_label, kernel_timing=kernel_note, launch_ms=round(float(dur_ms), 3), layer_call_ms=round(float(layer_ms), 3),
                        algorithmic_bytes_per_launch=int(alg_bytes), modmul_per_s=round(modmul_s, 1) if modmul_s else None)
        cpu = None
        if args.cpu_seconds > 0 and world == 1:      # (the CPU leg runs at N = 1 only: at N > 1 the host cores are busy driving N ranks)
            x0 = x_all[0].cpu().numpy().view(np.uint64).reshape(1, 28, 28, 2, E.k, E.n)
            cpu = cpu_baseline_reference(cfg, q, W, x0, host_cores(), evk=evk) or cpu_baseline(cfg, q, W, x0, args.cpu_seconds)
            cpu["value"] = round(cpu["value"], 6); cpu["mac_per_s"] = round(cpu["mac_per_s"], 1)
            try:            # configs[0] measured in full in the build container (not extrapolated): oracle/make_c1.py
                c1f = json.load(open(c1_path))
                if cfg_name == "tiny4096":
                    cpu["c1_in_full"] = dict(images=len(c1f["images"]), images_per_s=c1f.get("images_per_s_adjusted", c1f["images_per_s"]),
                                             total_wall_s=c1f.get("total_wall_s_adjusted", c1f["total_wall_s"]), threads=c1f["ref_threads"], note=c1f.get("note"),
                                             where="build container (8 cores), the compiled reference on 32 images: tests/golden/c1_tiny4096_t32.json")
            except Exception:
                pass
        value = B * world * steps / elapsed
        result = {
            "metric": "encrypted images/sec", "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": f"synthetic ({D} distinct MNIST-like encrypted images per GPU tiled to the batch" + ("" if window == B else f", resident as a {window}-image window") + f"; trained weights from {model}.h5)",
            "config": {"workload": f"{model}.h5 n={cfg['n']} k={cfg['k']} t=2^{cfg['t'].bit_length() - 1} batch={B}/GPU chunk={C}" + (f" (dense layers: {C * G})" if G > 1 else "") + f" ({cfg_name}, BASELINE configs)",
                       "mode": args.mode + ("+conv/pool folding" if want_fuse else ""), "parallelism": f"image-sharded x{world}, RCCL weight broadcast"},
            "ms_per_layer": ms_per_layer,
            "mac_kernel_per_layer": {pl[1]: ("mfma_mac2w_kernel (int8 limb GEMM, CRC_NTTL)" if pl[3].get("w_form") == ca.NTTL else
                                             "mfma_conv1_kernel (one-channel convolution on the matrix cores, CRC_NTTL1)" if pl[3].get("w_form") == ca.NTTL1 else
                                             (pl[3]["stream_kernel"] + ", streamed weights (coefficient-form plaintexts lifted + transformed a filter tile at a time)" if pl[3].get("streamed") else
                                              "mac3_kernel (v_mad_u64_u32, CRC_NTTP)") +
                                             (f" [{pl[3]['limb_skipped']}]" if pl[3].get("limb_skipped") else "")) for pl in net.plan if pl[0] in ("conv", "fc")},
            "mfma_useful_frac_per_layer": {pl[1]: round(1.0 / limb_exec_over_useful(pl[0], pl[2], C * G if li >= net.split else C, int(np.prod(pl[5]))), 4)
                                           for li, pl in enumerate(net.plan) if pl[0] in ("conv", "fc") and pl[3].get("w_form") == ca.NTTL},
            "reference_layer_structure": unfused, "roofline": roofline, "cpu_baseline": cpu,
            "check": {"tiled_outputs_identical": bool(ok_tile), "predictions_match_plain_model": f"{preds_ok}/{D}", "max_logit_abs_err": round(max_err, 6),
                      "noise_budget_bits": budgets, "ranks_verified": f"{ranks_ok}/{world}", "golden_match": gold_ok, "golden": gold_name, "c1_images_match_reference": c1_ok, "all_ok": bool(all_ok)},
            "setup_s": round(setup_s, 1), "weight_broadcast": bcast, "weight_broadcast_s": bcast["seconds"] if bcast else 0.0, "weight_bytes": int(net.weight_bytes),
        }
    # the same workload through the C++ host classes (crcnn_amd/host: the drop-in for the reference's Layer / Network / CnnBuilder) is measured by main() once this
    # engine has given its memory back: the distinct encrypted images go to a scratch file
    # (by default for the configurations below 60 GiB of encoded weights, where building the network a second time costs seconds; --host-cpp 2 for any: at WoPad 16384 the
    # C++ classes build their 182 GiB of limb weights tile by tile and chunk on two levels exactly like the Python twin, ~25 s of setup)
    if result is not None and full and world == 1 and args.host_cpp and (est_w_gib < 60 or args.host_cpp == 2) and q == ca.default_coeff_modulus_128(cfg["n"])[:cfg["k"]]:
        import tempfile
        hd = tempfile.mkdtemp(prefix="crc_host_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        x_all[:D].cpu().numpy().tofile(os.path.join(hd, "inputs.u64"))
        result["_host_job"] = dict(dir=hd, model=model, n=cfg["n"], k=cfg["k"], t=cfg["t"], distinct=D, batch=B, chunk=C, group=G, golden=GOLDEN_FOR.get(cfg_name),
                                   golden_input_ok=bool(gold_ok is not None), python_images_per_s=result["value"])
    # give everything back before a second workload
    del net, x_all, out_all, outs, keep[:]
    E.sync(); E.close()
    torch.cuda.empty_cache()
    return result, bool(all_ok)


def host_cpp_leg(line, job, args):
    """north_star's "C++ host code calls hand-written HIP kernels" as a measured path: crcnn_amd/lib/bench_host builds the same network with the C++ CnnBuilder, fuses it,
    and times Network::forward over the same encrypted images in the same chunks.  The check fails when its 10 output ciphertexts of image 0 are not the golden's
    (= this run's) or its images/s differ from the Python twin's by more than 3 %"""
    import hashlib
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "crcnn_amd", "lib", "bench_host")
    h5 = os.path.join(ROOT, "tests", "golden", "models", job["model"] + ".h5")
    out0 = os.path.join(job["dir"], "out0.u64")
    steps = max(1, min(args.steps, args.host_cpp_steps))
    res = dict(binary="crcnn_amd/lib/bench_host (crcnn_amd/host/bench_host.cpp)")
    try:
        if not os.path.exists(exe):
            raise RuntimeError("crcnn_amd/lib/bench_host has not been built")
        p = subprocess.run([exe, job["model"], h5, str(job["n"]), str(job["k"]), str(job["t"]), os.path.join(job["dir"], "inputs.u64"), str(job["distinct"]), str(job["batch"]),
                            str(job["chunk"]), str(steps), out0, str(job["group"])], capture_output=True, text=True, timeout=900)
        if p.returncode != 0:
            raise RuntimeError(f"exit {p.returncode}: {p.stderr[-400:]}")
        r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        sha = hashlib.sha256(open(out0, "rb").read()).hexdigest()
        g = json.load(open(os.path.join(ROOT, "tests", "golden", f"net_{job['golden']}.json"))) if job["golden"] and job["golden_input_ok"] else None
        ratio = r["images_per_s"] / job["python_images_per_s"]
        res.update(images_per_s=r["images_per_s"], ms_per_image=r["ms_per_image"], steps=steps, chunk=r["chunk"], setup_s=r["setup_s"],
                   T_LAYER={f"T_LAYER_{i}": v for i, v in enumerate(r["T_LAYER_ms_per_image"])}, layers=r["layers"],
                   vs_python_twin=round(ratio, 4), within_3_percent=bool(abs(ratio - 1) <= 0.03),
                   golden_match=(sha == g["out_sha256"]) if g else None, out0_sha256=sha)
        ok = res["within_3_percent"] and res["golden_match"] is not False
    except Exception as ex:
        res["error"] = f"{type(ex).__name__}: {ex}"
        ok = False
    shutil.rmtree(job["dir"], ignore_errors=True)
    line["host_cpp"] = res
    line["check"]["host_cpp_ok"] = bool(ok)
    line["check"]["all_ok"] = bool(line["check"]["all_ok"] and ok)
    return ok


def main():
    args = parse()
    if args.launch_check:
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            sys.exit(self_launch(args))
        sys.exit(launch_check(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain `python bench.py --gpus N`: start the ranks ourselves
        sys.exit(self_launch(args))
    D_ = Dist(args)
    line, ok = run_config(args, D_, args.config, args.steps, args.warmup, batch=args.batch, chunk=args.chunk, full=True)
    if line is not None and line.get("_host_job"):
        ok = host_cpp_leg(line, line.pop("_host_job"), args) and ok
    also = args.also
    if also == "auto":
        # the default invocation also measures BASELINE configs[2] (ApproxPlainModel, n = 8192, k = 3) and configs[4]'s workload on one GPU (PlainModelWoPad, n = 16384,
        # k = 4: batch 96 -- 202 GiB of weights leave room for 6-image chunks)
        also = "approx8192,wopad16384" if args.config == "tiny4096" and args.batch is None else "none"
    if also != "none":
        for nm in also.split(","):
            b2 = args.also_batch or (96 if nm.startswith("wopad") else None)
            second, ok2 = run_config(args, D_, nm, 1 if nm.startswith("wopad") else args.also_steps, 0, batch=b2, full=False)
            ok = ok and ok2
            if line is not None:
                line.setdefault("also", []).append({k_: second[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "config", "data", "ms_per_layer", "mac_kernel_per_layer", "mfma_useful_frac_per_layer",
                                                                              "roofline", "cpu_baseline", "check", "setup_s", "weight_broadcast", "weight_bytes")})
    if line is not None:
        print(json.dumps(line), flush=True)
    if D_.world > 1:
        D_.dist.destroy_process_group()
    if not ok:
        sys.stderr.write("bench.py: a self-check FAILED (see \"check\"): the reported value is not a valid measurement\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
