#!/usr/bin/env python3
"""PROTOTYPE driver (round 6): CrCNN's conv2+pool2 layer (32ch 12x12 -> 64 x 4x4, 6x6 window stride 2, T = 1152) as a MULTI-MODULAR int8-MFMA GEMM
(tools/mfma_rns.hip: 16 products per modular multiply-add) beside the limb GEMM prototype (tools/mfma_mac.hip: 49) and the product's kernels on the same random NTT-form
operands: bit-for-bit comparison, then timing on the same box.
usage: bench_mfma_rns.py [n] [k] [B] [reps]        build first:  make -C tools"""
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
HERE = os.path.dirname(os.path.abspath(__file__))
M = ctypes.CDLL(os.path.join(HERE, "libmfma_mac.so")); R = ctypes.CDLL(os.path.join(HERE, "libmfma_rns.so"))
for lib, names in ((M, ("mm_xp_bytes", "mm_wp_bytes", "mm_ys_bytes")), (R, ("rr_xp_bytes", "rr_wp_bytes", "rr_ys_bytes", "rr_mod_bytes"))):
    for f in names:
        getattr(lib, f).restype = ctypes.c_size_t
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
for i in range(k):                                     # edge values: 0, q-1, q/2 +- 1 in the first rows
    x[i, 0] = 0; x[i, 1] = q[i] - 1; x[i, 2] = q[i] // 2; x[i, 3] = q[i] // 2 + 1; w[i, 0] = q[i] - 1; w[i, 1] = q[i] // 2 + 1; w[i, 2] = q[i] // 2
# the worst case of the exactness bound in slot 5: EVERY operand of image 0 and of filter 0 at (q-1)/2, so that output (0, filter 0, every pixel) is T ((q-1)/2)^2
for i in range(k):
    x[i:in_cts * 2 * k:k, 5] = q[i] // 2
    w[i:T * k:k, 5] = q[i] // 2
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


def ck(rc, what):
    if rc:
        raise RuntimeError(f"{what}: {rc}")


qa = (ctypes.c_uint64 * k)(*q)
t_ref = timed(ref)
res = {}
for name, lib, pre in (("limb (49 products)", M, "mm"), ("multi-modular (16 products)", R, "rr")):
    xp = torch.empty(getattr(lib, pre + "_xp_bytes")(n, k, B), dtype=torch.int8, device=dev)
    wp = torch.empty(getattr(lib, pre + "_wp_bytes")(n, k), dtype=torch.int8, device=dev)
    ys = torch.empty(getattr(lib, pre + "_ys_bytes")(n, k, B) // 8, dtype=torch.int64, device=dev)
    y = torch.empty_like(y_ref)
    mods = torch.empty(4096, dtype=torch.int64, device=dev)
    ck(getattr(lib, pre + "_pack_w")(VP(w.data_ptr()), VP(wp.data_ptr()), qa, n, k, VP(mods.data_ptr())), "pack_w")
    t_px = timed(lambda: ck(getattr(lib, pre + "_pack_x")(VP(x.data_ptr()), VP(xp.data_ptr()), n, k, B, VP(mods.data_ptr())), "pack_x"), 1)
    conv = lambda mode=0: ck(getattr(lib, pre + "_conv")(VP(xp.data_ptr()), VP(wp.data_ptr()), VP(ys.data_ptr()), n, k, B, VP(mods.data_ptr()), mode), "conv")
    t_mm = timed(conv)
    t_nomfma = timed(lambda: conv(2))
    conv(0)
    ck(getattr(lib, pre + "_unpack_y")(VP(ys.data_ptr()), VP(y.data_ptr()), n, k, B), "unpack_y")
    torch.cuda.synchronize()
    same = bool(torch.equal(y, y_ref))
    res[name] = (t_mm, t_px, t_nomfma, same, int((y != y_ref).sum().item()) if not same else 0, (xp.numel() + wp.numel()) / 1e9)
    del xp, wp, ys, y
    torch.cuda.empty_cache()
modmul = B * P * NF * T * 2 * k * n
print(f"n={n} k={k} B={B} (conv2+pool2 of PlainModelTiny, T = {T}): {modmul / 1e12:.3f} T modular multiply-adds per launch")
print(f"  product, vector ALU (mac3_kernel):             {t_ref:8.2f} ms  {modmul / t_ref / 1e9:6.2f} T modmul/s")
for name, (t_mm, t_px, t_nm, same, bad, gb) in res.items():
    print(f"  prototype, {name:28s} {t_mm:8.2f} ms  {modmul / t_mm / 1e9:6.2f} T modmul/s  bit-identical to mac3_kernel: {same}{'' if same else f' ({bad} words differ)'}  | x pack {t_px:.2f} ms | "
          f"without its MFMAs {t_nm:.2f} ms | operand tensors {gb:.2f} GB")
a, b = res["limb (49 products)"][0], res["multi-modular (16 products)"][0]
print(f"  multi-modular / limb prototype: {b / a:.3f}  (MFMA count 16 / 49 = {16 / 49:.3f}; operand planes 16 / 7 = {16 / 7:.2f})")
