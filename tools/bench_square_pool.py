#!/usr/bin/env python3
"""Square + relinearise + sum pooling: crc_square_relin_forms followed by crc_pool against crc_square_pool_relin_forms (one key switch per pooled ciphertext).
usage: python tools/bench_square_pool.py n k images   (CrCNN's act1 -> pool2: 50 channels of 5 x 5, 2 x 2 window, stride 1; NTT-resident in and out)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crcnn_amd as ca

n, k, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
q = ca.default_coeff_modulus_128(n)[:k]
E = ca.Engine(n, q, 1 << 30, device=0)
dev = torch.device("cuda", 0)
E.stream = torch.cuda.current_stream().cuda_stream or None
zd, xd, yd, xs, ys, xf, yf = 50, 5, 5, 1, 1, 2, 2
xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
cnt, ocnt = B * zd * xd * yd, B * zd * xo * yo
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.empty((cnt * 2 * k, n), dtype=torch.int64, device=dev)
for i in range(k):
    x[i::k] = torch.randint(0, q[i], (cnt * 2, n), dtype=torch.int64, device=dev, generator=g)
sk, pk = E.keygen(3); evk = E.upload(E.gen_evk(4, sk))
work = torch.empty(max(E.square_relin_work_bytes(cnt), E.square_pool_relin_work_bytes(B, zd, xd, yd, xs, ys, xf, yf)) // 8 + 64, dtype=torch.int64, device=dev)
r = torch.empty_like(x); p1 = torch.empty((ocnt * 2 * k, n), dtype=torch.int64, device=dev); p2 = torch.empty_like(p1)
def seq():
    E.square_relin(x, cnt, evk, r, work, in_form=ca.NTT, out_form=ca.NTT)
    E.pool(r, B, zd, xd, yd, xs, ys, xf, yf, None, ca.NTT, p1)
def fused():
    E.square_pool_relin(x, B, zd, xd, yd, xs, ys, xf, yf, evk, p2, work, in_form=ca.NTT, out_form=ca.NTT)
for name, fn in (("square_relin + pool", seq), ("square_pool_relin", fused), ("square_relin + pool", seq), ("square_pool_relin", fused)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    print(f"n={n} k={k} images={B} ({cnt} -> {ocnt} cts) {name}: {ms:.3f} ms  {ms / B:.3f} ms/image  {1e3 * ms / cnt:.2f} us per squared ct")
print("equal:", bool(torch.equal(p1, p2)))
