"""include/crcnn_hip.h's threading contract on hardware: ONE context, two host threads, two HIP streams, each thread with its own work buffers -- a convolution
(crc_conv2d) on one and Square + relinearise (crc_square_relin_forms) plus a device checksum (crc_checksum64, the one entry point with context scratch) on the
other, started together and repeated; every result must be the compiled reference's layer output (tests/golden/layers_n256_k2_t20.npz: ConvolutionalLayer and
SquareLayer of CrCNN itself) and every checksum the one a quiet context gives.  ctypes releases the GIL for the duration of a call, so the calls really overlap."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "layers_n256_k2_t20.npz")


def test_two_host_threads_two_streams_one_context():
    import ctypes
    import torch
    import crcnn_amd as ca
    from crcnn_amd import binding
    g = dict(np.load(G))
    E = ca.Engine(int(g["n"]), [int(x) for x in g["q"]], int(g["t"]), device=0)
    L = E.L
    zd, xd, yd, xs, ys, xf, yf, nf = [int(v) for v in g["dims"][:8]]
    xo, yo = (xd - xf) // xs + 1, (yd - yf) // ys + 1
    # parameters (made on the default stream, before the threads start)
    pl, _ = E.encode(np.asarray(g["conv_w"], dtype=np.float32)); d_w = E.alloc(len(pl) * E.k * E.n * 8); E.plain_to_ntt(E.upload(pl), len(pl), d_w)
    pb, _ = E.encode(np.asarray(g["conv_b"], dtype=np.float32)); d_b = E.alloc(len(pb) * E.k * E.n * 8); E.plain_to_delta(E.upload(pb), len(pb), ca.COEFF, d_b)
    B = 4
    x = np.ascontiguousarray(np.repeat(g["x"][None], B, axis=0))
    d_x = E.upload(x); d_evk = E.upload(g["evk"])
    cts = B * zd * xd * yd
    E.sync()
    streams = [torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)]
    reps, errors, results = 12, [], {"conv": [], "square": [], "sum": []}
    quiet = np.zeros(2, dtype=np.uint64)
    binding._chk(L.crc_checksum64(E.c, E.p(d_x), x.size, quiet.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), None), "crc_checksum64")
    start = threading.Barrier(2)

    def conv_thread():
        try:
            st = ctypes.c_void_p(streams[0].cuda_stream)
            d_y = E.alloc(B * nf * xo * yo * 2 * E.k * E.n * 8); d_work = E.alloc(E.conv2d_work_bytes(B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF))
            start.wait()
            for _ in range(reps):
                binding._chk(L.crc_conv2d(E.c, E.p(d_x), E.p(d_w), E.p(d_b), B, zd, xd, yd, xs, ys, xf, yf, nf, ca.COEFF, ca.COEFF, E.p(d_y), E.p(d_work), st), "crc_conv2d")
                out = np.empty((B, nf, xo, yo, 2, E.k, E.n), dtype=np.uint64)
                binding._chk(L.crc_memcpy_d2h(E.c, out.ctypes.data, E.p(d_y), out.nbytes, st), "crc_memcpy_d2h"); binding._chk(L.crc_stream_sync(E.c, st), "crc_stream_sync")
                results["conv"].append(out)
        except Exception as ex:                  # noqa: BLE001 -- reported by the main thread
            errors.append(ex)

    def square_thread():
        try:
            st = ctypes.c_void_p(streams[1].cuda_stream)
            d_y = E.alloc(x.nbytes); d_work = E.alloc(E.square_relin_work_bytes(cts))
            start.wait()
            for _ in range(reps):
                binding._chk(L.crc_square_relin_forms(E.c, E.p(d_x), ca.COEFF, cts, E.p(d_evk), 16, E.p(d_y), ca.COEFF, E.p(d_work), st), "crc_square_relin_forms")
                cs = np.zeros(2, dtype=np.uint64)
                binding._chk(L.crc_checksum64(E.c, E.p(d_x), x.size, cs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), st), "crc_checksum64")
                out = np.empty(x.shape, dtype=np.uint64)
                binding._chk(L.crc_memcpy_d2h(E.c, out.ctypes.data, E.p(d_y), out.nbytes, st), "crc_memcpy_d2h"); binding._chk(L.crc_stream_sync(E.c, st), "crc_stream_sync")
                results["square"].append(out); results["sum"].append(cs)
        except Exception as ex:                  # noqa: BLE001
            errors.append(ex)

    th = [threading.Thread(target=conv_thread), threading.Thread(target=square_thread)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    assert len(results["conv"]) == reps and len(results["square"]) == reps
    for r in results["conv"]:
        for b in range(B):
            assert np.array_equal(r[b], g["ref_conv"])
    for r in results["square"]:
        for b in range(B):
            assert np.array_equal(r[b], g["ref_square"])
    for cs in results["sum"]:
        assert np.array_equal(cs, quiet)
    E.close()
