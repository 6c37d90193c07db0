"""conv1 kernel (CRC_NTTL1) at full size: parity against the vector-ALU kernel + timing.  usage: check_conv1.py n k B [tiny|approx]"""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import crcnn_amd as ca

n, k, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
shape = (28, 28, 2, 2, 6, 6, 32) if (len(sys.argv) < 5 or sys.argv[4] == "tiny") else (28, 28, 2, 2, 7, 7, 20)
xd, yd, xs, ys, xf, yf, nf = shape
q = ca.default_coeff_modulus_128(n)[:k] if k <= len(ca.default_coeff_modulus_128(n)) else None
E = ca.Engine(n, q, 1 << 32, device=0)
print("q bits", [int(v).bit_length() for v in E.q])
rng = np.random.default_rng(1)
def rows(r):
    out = np.empty((r, E.k, E.n), dtype=np.uint64)
    for i, qq in enumerate(E.q):
        out[:, i] = rng.integers(0, qq, size=(r, E.n), dtype=np.uint64)
    return out
xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
x1 = rows(xd * yd * 2)
w = rows(nf * xf * yf); b = rows(nf)
d_x1, d_w, d_b = E.upload(x1), E.upload(w), E.upload(b)
d_x = E.alloc(B * x1.nbytes)                      # the same image B times (replicated on the device)
for bb in range(B):
    E.L.crc_memcpy_d2d(E.c, E.p(d_x) + bb * x1.nbytes, E.p(d_x1), x1.nbytes, E.stream)
E.sync()
rows_y = B * nf * xo * yo * 2 * E.k
d_y0 = E.alloc(rows_y * E.n * 8)
d_work = E.alloc(E.conv2d_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT))
E.conv2d(d_x, d_w, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTT, d_y0, d_work); E.sync()
t0 = time.perf_counter(); E.conv2d(d_x, d_w, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTT, d_y0, d_work); E.sync()
print("vector ALU: %.2f ms/image" % ((time.perf_counter() - t0) * 1e3 / B))
d_wl = E.alloc(E.limb_conv1_weights_bytes()); E.limb_conv1_pack_weights(d_w, nf, xf, yf, d_wl)
nb = E.limb_tensor_bytes(B, nf, xo, yo)
d_y = E.alloc(max(rows_y * E.n * 8, nb))
for fout in (ca.NTT, ca.NTTLC):
    d_wk = E.alloc(E.conv2d_forms_work_bytes(B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, ca.NTTL1, fout))
    E.conv2d(d_x, d_wl, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, fout, d_y, d_wk, w_form=ca.NTTL1); E.sync()
    t0 = time.perf_counter()
    E.conv2d(d_x, d_wl, d_b, B, 1, xd, yd, xs, ys, xf, yf, nf, ca.NTT, fout, d_y, d_wk, w_form=ca.NTTL1); E.sync()
    print("conv1 kernel out_form %d: %.2f ms/image" % (fout, (time.perf_counter() - t0) * 1e3 / B))
    if fout == ca.NTT:
        per = rows_y // B
        bad = [bb for bb in range(B) if not np.array_equal(E.download(E.p(d_y) + bb * per * E.n * 8, (per, E.n)), E.download(E.p(d_y0) + bb * per * E.n * 8, (per, E.n)))]
        print("NTT out: images that differ:", bad)
    else:
        d_ref = E.alloc(nb); E.L.crc_memset(E.c, E.p(d_ref), 0, nb, E.stream)
        E.limb_pack_tensor(d_y0, ca.NTT, B, nf, xo, yo, d_ref); E.sync()
        A = E.download(d_y, (E.k * E.n, B, nb // (E.k * E.n * B * 8))); R = E.download(d_ref, (E.k * E.n, B, nb // (E.k * E.n * B * 8)))
        bad = [bb for bb in range(B) if not np.array_equal(A[:, bb], R[:, bb])]
        print("limb out: images that differ:", bad, "slots", np.unique(np.nonzero((A != R).any(axis=(1, 2)))[0])[:20])
    del d_wk
