#!/usr/bin/env python3
"""PROTOTYPE driver (VERDICT r1 task 6): CrCNN's conv2+pool2 layer (32ch 12x12 -> 64 x 4x4, 6x6 window stride 2, T = 1152) as an int8-MFMA limb
GEMM (tools/mfma_mac.hip) against the product's mac3_kernel on the same random NTT-form operands: bit-for-bit comparison, then timing.
usage: bench_mfma.py [n] [k] [B] [reps]        build first:  make -C tools"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crcnn_amd as ca

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 24
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
q = ca.default_coeff_modulus_128(4096 if n <= 4096 else n)[:k]
E = ca.Engine(n, q, 1 << 32, device=0)
dev = torch.device("cuda", 0)
M = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_mac.so"))
for f in ("mm_xp_bytes", "mm_wp_bytes", "mm_ys_bytes"):
    getattr(M, f).restype = ctypes.c_size_t
VP = ctypes.c_void_p
ZD, XD, WF, NF, P = 32, 12, 6, 64, 16
T = ZD * WF * WF
in_cts = ZD * XD * XD


def rand_rows(rows):
    t = torch.empty((rows, n), dtype=torch.int64, device=dev)
    for i in range(k):
        t[i::k] = torch.randint(0, q[i], ((rows + k - 1 - i) // k, n), dtype=torch.int64, device=dev)
    return t


torch.manual_seed(1)
x = rand_rows(B * in_cts * 2 * k)                      # [B][in_cts][2][k][n]
w = rand_rows(NF * T * k)                              # [NF][ZD][6][6][k][n]
# edge values: 0, q-1, q/2 +- 1 in the first rows
for i in range(k):
    x[i, 0] = 0; x[i, 1] = q[i] - 1; x[i, 2] = q[i] // 2; x[i, 3] = q[i] // 2 + 1; w[i, 0] = q[i] - 1; w[i, 1] = q[i] // 2 + 1
y_ref = torch.empty((B * NF * P * 2 * k, n), dtype=torch.int64, device=dev)
work = torch.empty(E.conv2d_work_bytes(B, ZD, XD, XD, 2, 2, WF, WF, NF, ca.NTT) // 8 + 64, dtype=torch.int64, device=dev)


def ref():
    E.conv2d(x, w, None, B, ZD, XD, XD, 2, 2, WF, WF, NF, ca.NTT, ca.NTT, y_ref, work)


def timed(fn, r=reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(r):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / r


t_ref = timed(ref)
xp = torch.empty(M.mm_xp_bytes(n, k, B), dtype=torch.int8, device=dev)
wp = torch.empty(M.mm_wp_bytes(n, k), dtype=torch.int8, device=dev)
ys = torch.empty(M.mm_ys_bytes(n, k, B) // 8, dtype=torch.int64, device=dev)
y = torch.empty_like(y_ref)
mods = torch.empty(64, dtype=torch.int64, device=dev)
qa = (ctypes.c_uint64 * k)(*q)
ck = lambda rc, what: (_ for _ in ()).throw(RuntimeError(f"{what}: {rc}")) if rc else None
ck(M.mm_pack_w(VP(w.data_ptr()), VP(wp.data_ptr()), qa, n, k, VP(mods.data_ptr())), "pack_w")
t_px = timed(lambda: ck(M.mm_pack_x(VP(x.data_ptr()), VP(xp.data_ptr()), n, k, B, VP(mods.data_ptr())), "pack_x"), 1)
conv = lambda mode=0: ck(M.mm_conv(VP(xp.data_ptr()), VP(wp.data_ptr()), VP(ys.data_ptr()), n, k, B, VP(mods.data_ptr()), mode), "conv")
abl = ""
if os.environ.get("MM_ABLATE"):
    abl = f" | ablation: no loads {timed(lambda: conv(1)):.2f} ms, no MFMA {timed(lambda: conv(2)):.2f} ms, no epilogue reduction {timed(lambda: conv(3)):.2f} ms"
t_mm = timed(conv)
if os.environ.get("MM_ZERO"):          # data dependence of the clock the chip holds (power): the same launch on all-zero limb tensors
    xp.zero_(); wp.zero_(); torch.cuda.synchronize()
    abl += f" | all-zero operands {timed(conv):.2f} ms"
    ck(M.mm_pack_w(VP(w.data_ptr()), VP(wp.data_ptr()), qa, n, k, VP(mods.data_ptr())), "pack_w")
    ck(M.mm_pack_x(VP(x.data_ptr()), VP(xp.data_ptr()), n, k, B, VP(mods.data_ptr())), "pack_x"); conv(); torch.cuda.synchronize()
ck(M.mm_unpack_y(VP(ys.data_ptr()), VP(y.data_ptr()), n, k, B), "unpack_y")
torch.cuda.synchronize()
if os.environ.get("MM_RESIDENT"):
    t_res = timed(lambda: conv(4))
    y2 = torch.empty_like(y_ref)
    ck(M.mm_unpack_y(VP(ys.data_ptr()), VP(y2.data_ptr()), n, k, B), "unpack_y"); torch.cuda.synchronize()
    abl += f" | A-resident variant {t_res:.2f} ms, bit-identical: {bool(torch.equal(y2, y_ref))}"
    conv(0); torch.cuda.synchronize()
same = bool(torch.equal(y, y_ref))
bad = int((y != y_ref).sum().item()) if not same else 0
modmul = B * P * NF * T * 2 * k * n
print(f"n={n} k={k} B={B}: mac3_kernel {t_ref:.2f} ms ({modmul / t_ref / 1e9:.2f} T modmul/s) | mfma limb GEMM {t_mm:.2f} ms ({modmul / t_mm / 1e9:.2f} T modmul/s, "
      f"{49 * modmul / t_mm / 1e9:.0f} T int8 MAC/s) | x pack {t_px:.2f} ms | bit-identical: {same}{abl}" + ("" if same else f" ({bad} of {y.numel()} words differ)"))
