#!/usr/bin/env python3
"""Time the weight limb pack of a streamed dense layer: 64-filter limb tiles built from 8-filter canonical sub-tiles (crc_limb_pack_weights_tile), as netrun's
streamed layers do inside every forward.  usage: bench_pack.py [n] [k] [in_dim] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crcnn_amd as ca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
zd = int(sys.argv[3]) if len(sys.argv) > 3 else 3920
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
q = ca.default_coeff_modulus_128(n)[:k]
E = ca.Engine(n, q, 1 << 30, device=0)
dev = torch.device("cuda", 0)
ft, sub = 64, 8
w = torch.empty((sub * zd * k, n), dtype=torch.int64, device=dev)
for i in range(k):
    w[i::k] = torch.randint(0, q[i], (sub * zd, n), dtype=torch.int64, device=dev)
wl = torch.zeros(E.limb_weights_bytes(ft, zd, 1, 1), dtype=torch.int8, device=dev)
def run():
    for s0 in range(0, ft, sub):
        E.limb_pack_weights_tile(w, ft, s0, sub, zd, 1, 1, wl)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gb = (ft * zd * k * n * 8 + wl.numel()) / 1e9
print(f"n={n} k={k} in_dim={zd}: one 64-filter limb tile from 8 sub-tiles {ms:.2f} ms, {gb:.1f} GB read+written -> {gb / ms:.2f} TB/s   checksum {int(wl.view(torch.int64)[::100003].sum().item()) & 0xffffffff:x}")
