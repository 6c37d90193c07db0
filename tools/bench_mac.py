#!/usr/bin/env python3
"""Time one layer's ct x pt MAC launch on random residues (kernel-only view of the dominant kernel).
usage: python tools/bench_mac.py [conv2|conv1|fc3|conv1p|conv2p|aconv1|aconv2|afc3|f5] [B] [reps] [packed|limb|limbk]   (a*: ApproxPlainModel shapes; CRC_MAC2_CFG=16|8 forces a tile shape)
packed: CRC_NTTP operands on mac3_kernel; limb: the matrix-core path as a layer call (CRC_NTTP tensor in -> limb_pack_tensor + mfma_mac_kernel + result conversion -> CRC_NTT);
limbk: the same on a pre-converted limb tensor with a limb-form result (mfma_mac_kernel + slotmajor_to_limb only)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crcnn_amd as ca

layer = sys.argv[1] if len(sys.argv) > 1 else "conv2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, k, t = 4096, 2, 1 << 20
geo = {"conv2": dict(zd=32, xd=12, yd=12, xs=1, ys=1, xf=5, yf=5, nf=64), "conv1": dict(zd=1, xd=28, yd=28, xs=1, ys=1, xf=5, yf=5, nf=32),
       "conv2p": dict(zd=32, xd=12, yd=12, xs=2, ys=2, xf=6, yf=6, nf=64), "conv1p": dict(zd=1, xd=28, yd=28, xs=2, ys=2, xf=6, yf=6, nf=32),
       "aconv1": dict(zd=1, xd=28, yd=28, xs=2, ys=2, xf=5, yf=5, nf=20), "aconv2": dict(zd=20, xd=11, yd=11, xs=2, ys=2, xf=3, yf=3, nf=50),
       "afc3": dict(zd=800, xd=1, yd=1, xs=1, ys=1, xf=1, yf=1, nf=500), "f5": dict(zd=4, xd=28, yd=28, xs=1, ys=1, xf=5, yf=5, nf=5),
       "fc3": dict(zd=1024, xd=1, yd=1, xs=1, ys=1, xf=1, yf=1, nf=512), "fc4": dict(zd=512, xd=1, yd=1, xs=1, ys=1, xf=1, yf=1, nf=10)}[layer]
q = ca.default_coeff_modulus_128(n)[:k]
E = ca.Engine(n, q, t, device=0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(cts_rows):   # rows of n residues, row r mod k -> modulus
    out = torch.empty((cts_rows, n), dtype=torch.int64, device=dev)
    for i in range(k):
        out[i::k] = torch.randint(0, q[i], (len(range(i, cts_rows, k)), n), dtype=torch.int64, device=dev, generator=g)
    return out
a = geo
xo, yo = (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1
in_cts, out_cts, T = a["zd"] * a["xd"] * a["yd"], a["nf"] * xo * yo, a["zd"] * a["xf"] * a["yf"]
x = rnd(B * in_cts * 2 * k); w = rnd(a["nf"] * T * k); bias = rnd(a["nf"] * k)
y = torch.empty((B * out_cts * 2 * k, n), dtype=torch.int64, device=dev)
work = torch.empty(E.conv2d_work_bytes(B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], ca.NTT) // 8 + 64, dtype=torch.int64, device=dev)
MODE = sys.argv[4] if len(sys.argv) > 4 else ""
LIMB = MODE in ("limb", "limbk")
PACKED = os.environ.get("CRC_BENCH_PACKED") == "1" or MODE == "packed" or LIMB       # operands in the packed 28-bit limb form (CRC_NTTP)
if PACKED:
    E.stream = torch.cuda.current_stream().cuda_stream or None
    E.pack28(x, x.shape[0]); E.pack28(w, w.shape[0])
F_IO = ca.NTTP if PACKED else ca.NTT
if LIMB:
    E.pack28(w, w.shape[0], unpack=True)
    wl = torch.empty(E.limb_weights_bytes(a["nf"], a["zd"], a["xf"], a["yf"]), dtype=torch.int8, device=dev)
    E.limb_pack_weights(w, a["nf"], a["zd"], a["xf"], a["yf"], wl)
    work = torch.empty(E.conv2d_forms_work_bytes(B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], ca.NTTP, ca.NTTL, ca.NTT) // 8 + 64, dtype=torch.int64, device=dev)
    if MODE == "limbk":
        xl = torch.empty(E.limb_tensor_bytes(B, a["zd"], a["xd"], a["yd"]), dtype=torch.int8, device=dev)
        E.limb_pack_tensor(x, ca.NTTP, B, a["zd"], a["xd"], a["yd"], xl)
def run():
    if MODE == "limbk":
        E.conv2d(xl, wl, bias, B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], ca.NTTL, ca.NTTL, y, work, w_form=ca.NTTL)
    elif LIMB:
        E.conv2d(x, wl, bias, B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], ca.NTTP, ca.NTT, y, work, w_form=ca.NTTL)
    else:
        E.conv2d(x, w, bias, B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], F_IO, F_IO, y, work, w_form=F_IO)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
macs = B * out_cts * T
if PACKED and not LIMB:
    E.pack28(y, y.shape[0], unpack=True); torch.cuda.synchronize()
print(f"{layer} B={B}{' ' + (MODE or 'packed') if PACKED else ''}: {ms:.2f} ms/launch  {macs * 2 * k * n / ms / 1e9:.3f} T modmul/s  ({ms / B:.3f} ms/image)  checksum {int(y.view(-1)[::100003].sum().item()) & 0xffffffff:x}")
