#!/usr/bin/env python3
"""Time the row NTT kernels and the HBM-bound element-wise kernels on random residues.  usage: bench_ntt.py [n] [k] [cts]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crcnn_amd as ca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cts = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
q = ca.default_coeff_modulus_128(n)[:k]
E = ca.Engine(n, q, 1 << 20, device=0)
dev = torch.device("cuda", 0)
x = torch.randint(0, min(q), (cts * 2 * k, n), dtype=torch.int64, device=dev)
y = torch.randint(0, min(q), (cts * 2 * k, n), dtype=torch.int64, device=dev)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
nbytes = cts * 2 * k * n * 8
for name, fn, traffic in [("ntt_fwd", lambda: E.ntt_fwd(x, cts), 2 * nbytes), ("ntt_inv", lambda: E.ntt_inv(x, cts), 2 * nbytes),
                          ("add", lambda: E.add(x, y, cts), 3 * nbytes), ("multiply_plain_ntt(group=all)", lambda: E.multiply_plain_ntt(x, y, cts, cts), 2 * nbytes),
                          ("multiply_plain (ntt+mul+intt)", lambda: E.multiply_plain(x, y, cts, cts), 2 * nbytes)]:
    ms = timeit(fn)
    print(f"n={n} k={k} cts={cts} {name:32s} {ms:8.3f} ms  {traffic / ms / 1e6:8.1f} GB/s algorithmic  ({ms * 1e3 / cts:.2f} us/ct)")
