#!/usr/bin/env python3
"""Time Encryptor::encrypt on the device (crc_encrypt_dev[_forms]) on random plaintexts.  usage: bench_encrypt.py [n] [k] [cts] [form: 0 coefficient, 1 NTT]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crcnn_amd as ca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cts = int(sys.argv[3]) if len(sys.argv) > 3 else 128 * 784
form = int(sys.argv[4]) if len(sys.argv) > 4 else 0
q = ca.default_coeff_modulus_128(n)[:k]
t = 1 << 32
E = ca.Engine(n, q, t, device=0)
dev = torch.device("cuda", 0)
sk, pk = E.keygen(3)
d_pk = E.upload(pk)
pl = torch.randint(0, t, (cts, n), dtype=torch.int64, device=dev)
ct = torch.empty((cts * 2 * k, n), dtype=torch.int64, device=dev)
work = torch.empty((E.encrypt_dev_work_bytes(cts) + 7) // 8, dtype=torch.int64, device=dev)
def run(seed):
    if form and hasattr(E, "encrypt_dev_forms"): E.encrypt_dev_forms(d_pk, pl, cts, seed, ca.NTT, ct, work)
    else: E.encrypt_dev(d_pk, pl, cts, seed, ct, work)
run(1); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(3): run(2 + r)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"n={n} k={k} cts={cts} encrypt on the device ({'NTT' if form else 'coefficient'} form out): {ms:8.3f} ms  {ms * 1e3 / cts:.3f} us per ciphertext  {ms / (cts / 784):.3f} ms per image")
