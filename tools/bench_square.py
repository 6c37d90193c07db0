#!/usr/bin/env python3
"""Time Square + relinearise (crc_square_relin_forms) on random ciphertexts.  usage: bench_square.py [n] [k] [cts] [in_form] [out_form]
CRC_RELIN_UNFUSED=1 selects the separate relinearisation kernels.  CRC_BENCH_SQ_POOL=1: the layer pair Square + pooling with one key switch per pooled ciphertext
(crc_square_pool_relin_forms) on cts / 1250 images of CrCNN's act1 -> pool2 geometry (50 channels of 5 x 5, 2 x 2 window, stride 1); times are per SQUARED ciphertext."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crcnn_amd as ca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cts = int(sys.argv[3]) if len(sys.argv) > 3 else 1250
fin = int(sys.argv[4]) if len(sys.argv) > 4 else ca.NTT
fout = int(sys.argv[5]) if len(sys.argv) > 5 else ca.NTT
q = ca.default_coeff_modulus_128(max(n, 4096))[:k]
E = ca.Engine(n, q, 1 << 30, device=0)
dev = torch.device("cuda", 0)
sk, pk = E.keygen(1); evk = E.gen_evk(2, sk)
x = torch.empty((cts * 2 * k, n), dtype=torch.int64, device=dev)
for i in range(k):
    x[i::k] = torch.randint(0, q[i], (cts * 2, n), dtype=torch.int64, device=dev)
y = torch.empty_like(x)
d_evk = torch.from_numpy(evk.view(np.int64)).to(dev)
POOL = os.environ.get("CRC_BENCH_SQ_POOL") == "1"
B = cts // 1250
work = torch.empty(max(E.square_relin_work_bytes(cts), E.square_pool_relin_work_bytes(B, 50, 5, 5, 1, 1, 2, 2) if POOL else 0) // 8 + 64, dtype=torch.int64, device=dev)
if POOL:
    assert cts == B * 1250 and B >= 1, "with CRC_BENCH_SQ_POOL=1 cts must be a multiple of 1250 (whole images)"
def run():
    if POOL:
        E.square_pool_relin(x, B, 50, 5, 5, 1, 1, 2, 2, d_evk, y, work, 16, fin, fout)
    else:
        E.square_relin(x, cts, d_evk, y, work, 16, fin, fout)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"n={n} k={k} cts={cts}{' (pooled key switch: ' + str(B * 800) + ' out)' if POOL else ''} forms {fin}->{fout}: {ms:.3f} ms  {ms * 1e3 / cts:.2f} us/ct  checksum {int(y.view(-1)[::100003].sum().item()) & 0xffffffff:x}")
