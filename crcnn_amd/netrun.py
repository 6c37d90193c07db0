"""Python driver for whole CrCNN networks on the engine (used by bench.py and the tests).

Mirrors CnnBuilder::buildNetwork (CrCNN/src/cnnBuilder.cpp:108-179): the three hard-coded topologies, weights read from
the HDF5 model by dataset name, every float encoded with the fractional encoder, then `Network::forward`
(network.cpp:22-47) layer by layer over a batch of encrypted images.  All arithmetic happens in libcrcnn_hip.so; this
file only sequences C-ABI calls and owns device buffers.  The C++ twin of this logic is crcnn_amd/host/.
"""
import os

import numpy as np

from . import binding
from .binding import COEFF, NTT

# (kind, name, args)  -- argument order as in the reference's build*Layer calls
TOPOLOGIES = {
    # cnnBuilder.cpp:157-169
    "PlainModelTiny": [
        ("conv", "pool1_features.conv1", dict(xd=28, yd=28, zd=1, xs=1, ys=1, xf=5, yf=5, nf=32)),
        ("avgpool", "pool1", dict(xd=24, yd=24, zd=32, xs=2, ys=2, xf=2, yf=2)),
        ("conv", "pool2_features.conv2", dict(xd=12, yd=12, zd=32, xs=1, ys=1, xf=5, yf=5, nf=64)),
        ("avgpool", "pool2", dict(xd=8, yd=8, zd=64, xs=2, ys=2, xf=2, yf=2)),
        ("fc", "classifier.fc3", dict(in_dim=1024, out_dim=512)),
        ("fc", "classifier.fc4", dict(in_dim=512, out_dim=10)),
    ],
    # cnnBuilder.cpp:115-134
    "ApproxPlainModel": [
        ("conv", "pool1_features.conv1", dict(xd=28, yd=28, zd=1, xs=2, ys=2, xf=5, yf=5, nf=20)),
        ("avgpool", "pool1", dict(xd=12, yd=12, zd=20, xs=1, ys=1, xf=2, yf=2)),
        ("bn", "pool1_features.norm1", dict(ch=20)),
        ("conv", "pool2_features.conv2", dict(xd=11, yd=11, zd=20, xs=2, ys=2, xf=3, yf=3, nf=50)),
        ("square", "act1", dict()),
        ("avgpool", "pool2", dict(xd=5, yd=5, zd=50, xs=1, ys=1, xf=2, yf=2)),
        ("bn", "pool2_features.norm2", dict(ch=50)),
        ("fc", "classifier.fc3", dict(in_dim=800, out_dim=500)),
        ("fc", "classifier.fc4", dict(in_dim=500, out_dim=10)),
    ],
}
# cnnBuilder.cpp:136-155: same as Approx with sum pooling
TOPOLOGIES["PlainModelWoPad"] = [(("pool" if k == "avgpool" else k), n, a) for (k, n, a) in TOPOLOGIES["ApproxPlainModel"]]


def out_shape(kind, a, in_shape):
    if kind == "conv":
        return (a["nf"], (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1)
    if kind in ("pool", "avgpool"):
        return (a["zd"], (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1)
    if kind == "fc":
        return (1, a["out_dim"], 1)
    return in_shape


def layer_macs(kind, a):
    """ct x pt multiply-accumulates per image"""
    if kind == "conv":
        xo, yo = (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1
        return a["nf"] * xo * yo * a["zd"] * a["xf"] * a["yf"]
    if kind == "fc":
        return a["in_dim"] * a["out_dim"]
    return 0


def E_k_n_8(eng):
    return eng.k * eng.n * 8


class Network:
    """Encoded network resident in HBM.  `alloc(nbytes)` must return an object Engine.p() understands."""

    def __init__(self, eng, model, h5_path=None, weights=None, alloc=None, resident=True, encode_chunk=2048, dbc=16, d_evk=None, materialize=True, fuse_pool=None, limb=None):
        self.E, self.model, self.topo = eng, model, TOPOLOGIES[model]
        # conv / dense layers with long reductions run on the matrix cores (operand form CRC_NTTL, kernels_mfma.hip) unless CRC_MFMA=0
        self.limb = (os.environ.get("CRC_MFMA", "1") != "0") if limb is None else limb
        self._limbed = False
        # a conv / dense layer whose NTT-form weights (k rows per weight) would take more than this share of HBM keeps its weights as coefficient-form plaintexts
        # (ONE row per weight) and lifts + NTTs them a filter tile at a time inside every forward (SURVEY section 7's fall-back: PlainModelWoPad's fc3 at n = 16384,
        # k = 8 is 419 GB in NTT form, 52 GB as plaintexts).  Same ciphertexts; the price is k row transforms per weight and chunk.
        self.stream_share = float(os.environ.get("CRC_STREAM_SHARE", "0.75"))
        self.limb_reserve = 24 << 30          # HBM to leave free when a limb copy of the weights is made (activations + work space come later)
        self.alloc = alloc or eng.alloc
        self.resident = resident            # keep tensors NTT-resident between layers (bit-identical)
        self.dbc, self.d_evk = dbc, d_evk
        # fold a (sum/avg) pooling layer into the convolution in front of it when that reduces the MAC count (exact, SURVEY 8f-1+);
        # only in NTT-resident mode, where the intermediate tensor is not observable
        self.fuse_pool = resident if fuse_pool is None else (fuse_pool and resident)
        self.materialize = materialize      # False: only allocate parameter buffers (they are filled by an RCCL broadcast)
        self.param_bufs = []                # every parameter buffer in a deterministic order: (buffer, nbytes)
        self.weight_bytes = 0
        get = (lambda nm: np.asarray(weights[nm], dtype=np.float32)) if weights is not None else (lambda nm: binding.h5_read(h5_path, nm))
        self._keep = []
        mid_form = NTT if resident else COEFF
        shape = (1, 28, 28)
        form = COEFF
        self.plan = []
        for li, (kind, name, a) in enumerate(self.topo):
            p = {}
            last = li == len(self.topo) - 1
            if kind in ("conv", "fc"):
                out_form = COEFF if (last or not resident) else NTT
                wv = get(name + ".weight")
                if eng.device >= 0 and wv.size * E_k_n_8(eng) > self.stream_share * eng.mem_info()[1]:
                    p["w"] = None; p["streamed"] = True
                    p["plain"] = self._encode_plain(wv, encode_chunk)
                elif eng.device >= 0 and resident and self.limb_eligible(kind, a) and self._needs_tilewise(kind, a, wv.size):
                    # the matrix-core form of this layer's weights is built a filter tile at a time straight from the plaintexts (encode -> lift + NTT ->
                    # batch-norm fold -> limb tile), once the layer folding is known: the canonical NTT-form copy (202 GiB for PlainModelWoPad's fc3 at
                    # n = 16384, k = 4) never exists beside the limb copy (177 GiB)
                    nf_, zd_, xf_, yf_ = (a["nf"], a["zd"], a["xf"], a["yf"]) if kind == "conv" else (a["out_dim"], a["in_dim"], 1, 1)
                    nbytes = eng.limb_weights_bytes(nf_, zd_, xf_, yf_)
                    p["w"] = self.alloc(nbytes); p["w_form"] = binding.NTTL; p["tilewise"] = dict(wv=np.ascontiguousarray(wv, dtype=np.float32).reshape(nf_, -1), built=False)
                    self.weight_bytes += nbytes
                    self.param_bufs.append((p["w"], nbytes))
                    self._encode_chunk = encode_chunk
                else:
                    p["w"] = self._encode_ntt(wv, encode_chunk)
                p["b"] = self._delta(get(name + ".bias"), out_form)
                p["in_form"], p["out_form"] = form, out_form
                form = out_form
            elif kind in ("pool", "avgpool"):
                p["div"] = self._encode_ntt(np.array([1.0 / (a["xf"] * a["yf"])], dtype=np.float64), encode_chunk, dtype=np.float64) if kind == "avgpool" else None
                p["form"] = form
            elif kind == "bn":
                mean = get(name + ".running_mean"); var = get(name + ".running_var")
                invstd = np.float32(1.0 / np.sqrt(var.astype(np.float64) + 0.00001))       # cnnBuilder.cpp:100-102
                p["mean"] = self._delta(mean, form); p["invstd"] = self._encode_ntt(invstd, encode_chunk); p["form"] = form
            elif kind == "square":
                # crc_square_relin_forms takes and leaves NTT-resident tensors (one INTT inside feeds the BEHZ base extension)
                p["in_form"] = form
                p["out_form"] = COEFF if (last or not resident) else NTT
                form = p["out_form"]
            nshape = out_shape(kind, a, shape)
            self.plan.append((kind, name, a, p, shape, nshape))
            shape = nshape
        self.out_shape = shape
        self.out_form = form
        self._packed = False
        if self.fuse_pool:
            self.fuse()

    # ---- operand form of the MAC kernels (CRC_NTTP: 28-bit limb pairs): weights are packed once, and a conv / dense layer that feeds
    # another one hands its output over packed, so that no kernel has to split a residue again (+12 % on the conv / dense layers)
    @staticmethod
    def _geom(kind, a):
        """(zd, xd, yd, xs, ys, xf, yf, nf) of a conv / dense layer (a dense layer is the 1 x 1 convolution zd = in_dim, nf = out_dim)"""
        return (a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"]) if kind == "conv" else (a["in_dim"], 1, 1, 1, 1, 1, 1, a["out_dim"])

    def planned_form(self, kind, a, B=0):
        """the kernel (weight form) crc_plan_mac picks for the layer launched on B images (0: by shape alone) -- the policy lives behind the C ABI, shared with the C++ host
        classes: reductions of at least 8 steps of 32 channels with at least 32 rows per launch -> limb GEMM (CRC_NTTL), one-channel convolutions -> their own matrix-core
        kernel (CRC_NTTL1), everything else -> the vector-ALU kernel"""
        if kind not in ("conv", "fc"):
            return NTT
        wf = self.E.plan_mac(*self._geom(kind, a), B, matrix_cores=self.limb)
        if wf == binding.NTTL1 and os.environ.get("CRC_MFMA_CONV1", "1") == "0":
            wf = binding.NTTP
        return wf

    def limb_eligible(self, kind, a):
        return self.planned_form(kind, a, 0) == binding.NTTL

    def conv1_eligible(self, kind, a):
        return self.planned_form(kind, a, 0) == binding.NTTL1

    def _needs_tilewise(self, kind, a, count):
        """True when the canonical NTT-form weights and their limb copy (7/8 of the bytes, filters padded to 64) do not fit in what HBM has left, but the limb copy does"""
        E = self.E
        nf, zd, xf, yf = (a["nf"], a["zd"], a["xf"], a["yf"]) if kind == "conv" else (a["out_dim"], a["in_dim"], 1, 1)
        canon, limb = count * E_k_n_8(E), E.limb_weights_bytes(nf, zd, xf, yf)
        free = E.mem_info()[0]
        return canon + limb + self.limb_reserve > free and limb + self.limb_reserve + (8 << 30) <= free

    def _build_tilewise(self):
        """fill the limb weights of the tile-wise layers (after fuse(): a batch-norm layer folded into such a layer is applied to every tile before it is packed)"""
        E = self.E
        if not self.materialize:
            return
        rowb = E.k * E.n * 8
        for (kind, name, a, p, ishape, oshape) in self.plan:
            tw = p.get("tilewise")
            want = "folded" if p.get("bn_fold") else "plain"
            if not tw or tw["built"] == want:
                continue
            # (built "plain" before fuse() -- bench.py's pass over the reference layer structure -- it is built again with the fold; the bias is still the original)
            assert tw["built"] is False or want == "folded", "a batch-norm fold cannot be taken back"
            nf, zd, xf, yf = (a["nf"], a["zd"], a["xf"], a["yf"]) if kind == "conv" else (a["out_dim"], a["in_dim"], 1, 1)
            T = zd * xf * yf
            wv = tw["wv"]                                    # [nf][T] floats
            ft = int(max(1, min(nf, (4 << 30) // (T * rowb))))
            wt = self.alloc(ft * T * rowb)
            chunk = max(1, min(self._encode_chunk, ft * T))
            stage = self.alloc(chunk * E.n * 8); cstage = self.alloc(chunk * E.COMPACT_WORDS * 8)
            bn = p.get("bn_fold")
            if bn:
                fake = self.alloc(T * 2 * rowb); outc = self.alloc(ft * 2 * rowb)
                E.L.crc_memset(E.c, E.p(fake), 0, T * 2 * rowb, E.stream)
                for z in range(bn["ch"]):
                    for t in range(bn["per_ch"]):
                        E.L.crc_memcpy_d2d(E.c, E.p(fake) + (z * bn["per_ch"] + t) * 2 * rowb, E.p(bn["mean"]) + z * rowb, rowb, E.stream)
                wk = self.alloc(max(E.dense_work_bytes(1, T, ft, NTT), 256))
                bias = E.download(E.p(p["b"]), (nf, E.k, E.n))
                qv = np.array(E.q, dtype=np.uint64).reshape(1, E.k, 1)
            for f0 in range(0, nf, ft):
                fn = min(ft, nf - f0)
                vals = np.ascontiguousarray(wv[f0:f0 + fn]).reshape(-1)
                for o in range(0, vals.size, chunk):
                    cnt = E.encode_to_device(vals[o:o + chunk], stage, cstage)
                    E.plain_to_ntt(stage, cnt, E.p(wt) + o * rowb)
                    E.sync()
                if bn:
                    for f in range(fn):          # w'[f][z][tap] = w (*) s[z]
                        E.L.crc_multiply_plain_ntt(E.c, E.p(wt) + f * T * rowb, E.p(bn["invstd"]), T, bn["per_ch"], 1, E.stream)
                    E.dense(fake, wt, None, 1, T, fn, NTT, NTT, outc, wk)        # correction[f] = sum_t w'[f][t] (*) M[z(t)]
                    E.sync()
                    corr = E.download(E.p(outc), (fn, 2, E.k, E.n))[:, 0]
                    bias[f0:f0 + fn] = (bias[f0:f0 + fn] + (qv - corr)) % qv
                E.limb_pack_weights_tile(wt, nf, f0, fn, zd, xf, yf, p["w"])
                E.sync()
            if bn:
                bias = np.ascontiguousarray(bias)
                E.L.crc_memcpy_h2d(E.c, E.p(p["b"]), bias.ctypes.data, bias.nbytes, E.stream)
                E.sync()
                for b_ in (fake, outc, wk):
                    self._free(b_)
            self._free(wt); self._free(stage); self._free(cstage)
            tw["built"] = want

    def _limb_operands(self, B=None, B_tail=None, split=None):
        """conv / dense weights of the eligible layers -> limb form (the canonical copy is dropped: call after fuse()); a limb layer that feeds a dense
        limb layer hands its tensor over in limb form"""
        E = self.E
        if self._limbed or not self.limb:
            return
        assert not self._packed
        self._build_tilewise()
        for idx, (kind, name, a, p, ishape, oshape) in enumerate(self.plan):
            if self.conv1_eligible(kind, a) and not p.get("streamed"):
                nbytes = E.limb_conv1_weights_bytes()
                wl = self.alloc(nbytes)
                E.limb_conv1_pack_weights(p["w"], a["nf"], a["xf"], a["yf"], wl)
                E.sync()
                self.weight_bytes += nbytes - a["nf"] * a["xf"] * a["yf"] * E.k * E.n * 8
                self._free(p["w"])
                p["w"], p["w_form"] = wl, binding.NTTL1
                continue
            if not self.limb_eligible(kind, a) or p.get("streamed") or p.get("tilewise"):
                continue
            # (crc_plan_mac: with fewer than half a 64-row tile of rows = images x 2 polys x output pixels per launch the vector-ALU kernel is faster)
            Bl = B if (B_tail is None or split is None or idx < split) else B_tail          # images this layer is launched on
            if Bl is not None and self.planned_form(kind, a, Bl) != binding.NTTL:
                p["limb_skipped"] = "fewer than 32 rows per launch"
                continue
            nf, zd, xf, yf = (a["nf"], a["zd"], a["xf"], a["yf"]) if kind == "conv" else (a["out_dim"], a["in_dim"], 1, 1)
            nbytes = E.limb_weights_bytes(nf, zd, xf, yf)
            # the limb copy is built beside the canonical one: a layer whose two copies do not fit in HBM stays on the vector-ALU kernel
            if E.mem_info()[0] < nbytes + self.limb_reserve:
                p["limb_skipped"] = f"limb copy of {nbytes >> 30} GiB + {self.limb_reserve >> 30} GiB reserve > {E.mem_info()[0] >> 30} GiB free"
                continue
            wl = self.alloc(nbytes)
            E.limb_pack_weights(p["w"], nf, zd, xf, yf, wl)
            E.sync()
            self.weight_bytes += nbytes - nf * zd * xf * yf * E.k * E.n * 8
            self._free(p["w"])
            p["w"], p["w_form"] = wl, binding.NTTL
        for idx, (kind, name, a, p, ishape, oshape) in enumerate(self.plan):
            nxt = self.plan[idx + 1] if idx + 1 < len(self.plan) else None
            if p.get("w_form") == binding.NTTL and nxt and nxt[0] == "fc" and nxt[3].get("w_form") == binding.NTTL and p["out_form"] == NTT and nxt[3]["in_form"] == NTT:
                p["out_form"] = nxt[3]["in_form"] = binding.NTTL
            # a one-channel convolution in front of a matrix-core convolution writes that layer's limb tensor itself
            if p.get("w_form") == binding.NTTL1 and nxt and nxt[0] == "conv" and nxt[3].get("w_form") == binding.NTTL and p["out_form"] == NTT and nxt[3]["in_form"] == NTT:
                p["out_form"], nxt[3]["in_form"] = binding.NTTLC, binding.NTTL
        self._limbed = True

    def _free(self, buf):
        """give a parameter buffer back (torch tensors handed out by bench.py's allocator are dropped from its keep list through `release`)"""
        self.param_bufs = [(b_, n_) for (b_, n_) in self.param_bufs if b_ is not buf]
        if getattr(self, "release", None):
            self.release(buf)
        elif hasattr(buf, "free"):
            buf.free()

    def _pack_operands(self, unpack=False):
        """called by prepare() (pack) and by fuse() (unpack: the folding kernels work on canonical residues)"""
        E = self.E
        if unpack:
            assert not self._limbed, "fuse() must run before the weights are converted to limb form"
        if max(int(q).bit_length() for q in E.q) > 55 or self._packed == (not unpack):
            return
        for idx, (kind, name, a, p, ishape, oshape) in enumerate(self.plan):
            if kind in ("pool", "avgpool"):               # a pooling layer in front of a conv / dense layer writes the packed form too
                nxt = self.plan[idx + 1] if idx + 1 < len(self.plan) else None
                if nxt and nxt[0] in ("conv", "fc"):
                    if not unpack and p["form"] == NTT and nxt[3]["in_form"] == NTT:
                        p["form"] = nxt[3]["in_form"] = binding.NTTP
                    elif unpack and p["form"] == binding.NTTP:
                        p["form"] = nxt[3]["in_form"] = NTT
                continue
            if kind not in ("conv", "fc"):
                continue
            if p.get("w_form") not in (binding.NTTL, binding.NTTL1) and not p.get("streamed"):
                rows = (a["nf"] * a["zd"] * a["xf"] * a["yf"] if kind == "conv" else a["in_dim"] * a["out_dim"]) * E.k
                E.pack28(p["w"], rows, unpack=unpack)
                p["w_form"] = NTT if unpack else binding.NTTP
            nxt = self.plan[idx + 1] if idx + 1 < len(self.plan) else None
            if nxt and nxt[0] in ("conv", "fc"):
                if not unpack and p["out_form"] == NTT and nxt[3]["in_form"] == NTT:
                    p["out_form"] = nxt[3]["in_form"] = binding.NTTP
                elif unpack and p["out_form"] == binding.NTTP:
                    p["out_form"] = nxt[3]["in_form"] = NTT
        E.sync()
        self._packed = not unpack

    # ---- conv + pool fusion (crc_conv2d_fold_pool)
    def fuse(self):
        """fold pooling layers into the convolutions in front of them (exact); call prepare() again afterwards"""
        E = self.E
        if self.materialize:
            self._pack_operands(unpack=True)
        plan, i = [], 0
        while i < len(self.plan):
            kind, name, a, p, ishape, oshape = self.plan[i]
            nxt = self.plan[i + 1] if i + 1 < len(self.plan) else None
            if kind == "conv" and not p.get("streamed") and not p.get("tilewise") and nxt and nxt[0] in ("pool", "avgpool") and p["out_form"] == NTT and nxt[3]["form"] == NTT:
                pa = nxt[2]
                xf2, yf2 = (pa["xf"] - 1) * a["xs"] + a["xf"], (pa["yf"] - 1) * a["ys"] + a["yf"]
                xs2, ys2 = a["xs"] * pa["xs"], a["ys"] * pa["ys"]
                xo2, yo2 = (a["xd"] - xf2) // xs2 + 1, (a["yd"] - yf2) // ys2 + 1
                macs_sep = layer_macs("conv", a)
                macs_fused = a["nf"] * xo2 * yo2 * a["zd"] * xf2 * yf2
                same_shape = (a["nf"], xo2, yo2) == tuple(nxt[5])
                # (crc_plan_fold_pool: the cost model behind the C ABI, shared with the C++ host classes -- folding wins whenever it removes MACs and narrowly for
                # CrCNN's stride-1 pools)
                if same_shape and E.plan_fold_pool(a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], pa["xs"], pa["ys"], pa["xf"], pa["yf"]):
                    cnt = a["nf"] * a["zd"] * xf2 * yf2
                    w2 = self.alloc(cnt * E.k * E.n * 8); b2 = self.alloc(a["nf"] * E.k * E.n * 8)
                    if self.materialize:
                        E.conv2d_fold_pool(p["w"], p["b"], nxt[3]["div"], a["nf"], a["zd"], a["xf"], a["yf"], a["xs"], a["ys"], pa["xf"], pa["yf"], w2, b2)
                        E.sync()
                    # the fused parameters replace the originals in the broadcast list
                    self.param_bufs = [(b_, n_) for (b_, n_) in self.param_bufs if b_ is not p["w"] and b_ is not p["b"]]
                    self.param_bufs += [(w2, cnt * E.k * E.n * 8), (b2, a["nf"] * E.k * E.n * 8)]
                    self.weight_bytes += (cnt - a["nf"] * a["zd"] * a["xf"] * a["yf"]) * E.k * E.n * 8
                    a2 = dict(a, xs=xs2, ys=ys2, xf=xf2, yf=yf2)
                    p2 = dict(p, w=w2, b=b2, fused=(name, nxt[1]), macs_separate=macs_sep, macs=macs_fused)
                    p["w"] = p["b"] = None            # drop the unfused copies
                    plan.append(("conv", name + "+" + nxt[1], a2, p2, ishape, nxt[5]))
                    i += 2
                    continue
            plan.append(self.plan[i]); i += 1
        self.plan = plan
        self._pair_square_pool()
        self._fold_batchnorm()

    def _pair_square_pool(self):
        """a Square layer with a (sum or average) pooling behind it: relinearisation is linear in the digit polynomials of c2, so the digits of a pooling window are
        added and ONE key switch serves the pooled ciphertext (crc_square_pool_relin_forms) -- the same ciphertexts, xo yo / (xd yd) of the key-switching work"""
        E = self.E
        plan, i = [], 0
        while i < len(self.plan):
            kind, name, a, p, ishape, oshape = self.plan[i]
            nxt = self.plan[i + 1] if i + 1 < len(self.plan) else None
            if (kind == "square" and nxt and nxt[0] in ("pool", "avgpool") and p["out_form"] == NTT and nxt[3]["form"] == NTT
                    and E.square_pool_relin_supported(nxt[2]["xf"], nxt[2]["yf"], self.dbc)):
                plan.append(("squarepool", name + "+" + nxt[1], dict(nxt[2]), dict(in_form=p["in_form"], out_form=NTT, div=nxt[3]["div"]), ishape, nxt[5]))
                i += 2
                continue
            plan.append(self.plan[i]); i += 1
        self.plan = plan

    # ---- batch-norm folding: bn (per-channel  s (*) (x - M), M on poly 0 only) followed by a conv / dense layer is that layer with
    # weights w (*) s[channel] and bias  B - sum_taps w (*) s (*) M  -- ring-linear over Z_q, so the ciphertexts are identical
    def _fold_batchnorm(self):
        E = self.E
        plan, i = [], 0
        rowb = E.k * E.n * 8
        while i < len(self.plan):
            kind, name, a, p, ishape, oshape = self.plan[i]
            nxt = self.plan[i + 1] if i + 1 < len(self.plan) else None
            if kind == "bn" and nxt and nxt[0] in ("conv", "fc") and p["form"] == NTT and nxt[3]["out_form"] == NTT and "fused" not in nxt[3] and not nxt[3].get("streamed"):
                nk, nname, na, np_, nish, nosh = nxt
                ch = ishape[0]
                if nk == "conv":
                    F, per_ch, T = na["nf"], na["xf"] * na["yf"], na["zd"] * na["xf"] * na["yf"]
                else:
                    F, per_ch, T = na["out_dim"], ishape[1] * ishape[2], na["in_dim"]
                assert T == ch * per_ch
                if np_.get("tilewise"):
                    # the fold is applied tile by tile when the limb weights are built (_build_tilewise)
                    np_["bn_fold"] = dict(mean=p["mean"], invstd=p["invstd"], ch=ch, per_ch=per_ch)
                elif self.materialize:
                    # w'[f][z][tap] = w (*) s[z]: one multiply_plain_ntt per output row (plaintext index = tap / per_ch)
                    for f in range(F):
                        E.L.crc_multiply_plain_ntt(E.c, E.p(np_["w"]) + f * T * rowb, E.p(p["invstd"]), T, per_ch, 1, E.stream)
                    # correction[f] = sum_t w'[f][t] (*) M[z(t)]: the dense kernel on one pseudo-image whose "ciphertexts" are (M[z(t)], 0)
                    fake = self.alloc(T * 2 * rowb); outc = self.alloc(F * 2 * rowb)
                    E.L.crc_memset(E.c, E.p(fake), 0, T * 2 * rowb, E.stream)
                    for z in range(ch):
                        for t in range(per_ch):
                            E.L.crc_memcpy_d2d(E.c, E.p(fake) + (z * per_ch + t) * 2 * rowb, E.p(p["mean"]) + z * rowb, rowb, E.stream)
                    wk = self.alloc(max(E.dense_work_bytes(1, T, F, NTT), 256))
                    E.dense(fake, np_["w"], None, 1, T, F, NTT, NTT, outc, wk)
                    E.sync()
                    corr = E.download(E.p(outc), (F, 2, E.k, E.n))[:, 0]
                    bias = E.download(E.p(np_["b"]), (F, E.k, E.n))
                    qv = np.array(E.q, dtype=np.uint64).reshape(1, E.k, 1)
                    newb = np.ascontiguousarray((bias + (qv - corr)) % qv)
                    E.L.crc_memcpy_h2d(E.c, E.p(np_["b"]), newb.ctypes.data, newb.nbytes, E.stream)
                    E.sync()
                    self._keep.append(newb)
                self.param_bufs = [(b_, n_) for (b_, n_) in self.param_bufs if b_ is not p["mean"] and b_ is not p["invstd"]]
                self.weight_bytes -= 2 * ch * rowb
                p2 = dict(np_, folded_bn=name)
                plan.append((nk, name + "+" + nname, na, p2, ishape, nosh))
                i += 2
                continue
            plan.append(self.plan[i]); i += 1
        self.plan = plan

    # ---- parameter upload
    def _encode_ntt(self, vals, chunk, dtype=np.float32):
        E = self.E
        vals = np.ascontiguousarray(np.asarray(vals, dtype=dtype).reshape(-1))
        out = self.alloc(vals.size * E.k * E.n * 8)
        self.weight_bytes += vals.size * E.k * E.n * 8
        self.param_bufs.append((out, vals.size * E.k * E.n * 8))
        if not self.materialize:
            return out
        stage = self.alloc(min(chunk, vals.size) * E.n * 8)
        cstage = self.alloc(min(chunk, vals.size) * E.COMPACT_WORDS * 8) if dtype == np.float32 else None
        for o in range(0, vals.size, chunk):
            if cstage is not None:          # 96 words per weight over PCIe, zero-extended on the device
                cnt = E.encode_to_device(vals[o:o + chunk], stage, cstage)
            else:
                pl, _ = E.encode(vals[o:o + chunk], dtype=dtype); cnt = len(pl)
                E.L.crc_memcpy_h2d(E.c, E.p(stage), pl.ctypes.data, pl.nbytes, E.stream)
            E.plain_to_ntt(stage, cnt, E.p(out) + o * E.k * E.n * 8)
            E.sync()
        self._free(stage)
        if cstage is not None:
            self._free(cstage)
        return out

    def _encode_plain(self, vals, chunk, dtype=np.float32):
        """coefficient-form plaintexts [count][n] resident in HBM (streamed layers)"""
        E = self.E
        vals = np.ascontiguousarray(np.asarray(vals, dtype=dtype).reshape(-1))
        out = self.alloc(vals.size * E.n * 8)
        self.weight_bytes += vals.size * E.n * 8
        self.param_bufs.append((out, vals.size * E.n * 8))
        if not self.materialize:
            return out
        cstage = self.alloc(min(chunk, vals.size) * E.COMPACT_WORDS * 8) if dtype == np.float32 else None
        for o in range(0, vals.size, chunk):
            if cstage is not None:
                E.encode_to_device(vals[o:o + chunk], E.p(out) + o * E.n * 8, cstage)
            else:
                pl, _ = E.encode(vals[o:o + chunk], dtype=dtype)
                E.L.crc_memcpy_h2d(E.c, E.p(out) + o * E.n * 8, pl.ctypes.data, pl.nbytes, E.stream)
                E.sync()
        if cstage is not None:
            self._free(cstage)
        return out

    def _delta(self, vals, form):
        E = self.E
        pl, _ = E.encode(np.asarray(vals, dtype=np.float32))
        out = self.alloc(len(pl) * E.k * E.n * 8)
        self.weight_bytes += len(pl) * E.k * E.n * 8
        self.param_bufs.append((out, len(pl) * E.k * E.n * 8))
        if not self.materialize:
            return out
        d_p = self.alloc(pl.nbytes)
        E.L.crc_memcpy_h2d(E.c, E.p(d_p), pl.ctypes.data, pl.nbytes, E.stream)
        E.plain_to_delta(d_p, len(pl), form, out)
        E.sync()
        return out

    # ---- sizes
    def ct_bytes(self):
        return 2 * self.E.k * self.E.n * 8

    def activation_cts(self):
        """ciphertexts per image at every layer boundary"""
        return [int(np.prod(s)) for (_, _, _, _, s, _) in self.plan] + [int(np.prod(self.out_shape))]

    def scratch_bytes(self, B):
        E = self.E
        acts = self.activation_cts()
        # two ping-pong activation buffers + the largest layer work buffer
        big = sorted(acts, reverse=True)
        need_act = (big[0] + big[1]) * B * self.ct_bytes()
        work = 0
        B0 = B
        for li, (kind, name, a, p, ishape, oshape) in enumerate(self.plan):
            B = B0 * getattr(self, "G", 1) if li >= getattr(self, "split", len(self.plan)) else B0
            if p.get("streamed"):
                g = self._stream_geometry(kind, a, B)
                if g["limb"]:
                    fin = binding.NTTL if p["in_form"] in (NTT, binding.NTTP, binding.NTTL) else p["in_form"]
                    work = max(work, E.conv2d_forms_work_bytes(B, g["zd"], g["xd"], g["yd"], g["xs"], g["ys"], g["xf"], g["yf"], g["ft"], fin, binding.NTTL, p["out_form"]))
                else:
                    work = max(work, E.conv2d_forms_work_bytes(B, g["zd"], g["xd"], g["yd"], g["xs"], g["ys"], g["xf"], g["yf"], g["ft"], p["in_form"], NTT, p["out_form"]))
                continue
            if kind == "conv":
                work = max(work, E.conv2d_forms_work_bytes(B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], p["in_form"], p.get("w_form", NTT), p["out_form"]))
            elif kind == "fc":
                work = max(work, E.conv2d_forms_work_bytes(B, a["in_dim"], 1, 1, 1, 1, 1, 1, a["out_dim"], p["in_form"], p.get("w_form", NTT), p["out_form"]))
            elif kind == "square":
                work = max(work, E.square_relin_work_bytes(B * int(np.prod(ishape)), self.dbc))
            elif kind == "squarepool":
                work = max(work, E.square_pool_relin_work_bytes(B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], self.dbc))
        return need_act, work

    def _stream_geometry(self, kind, a, B=None):
        """a streamed layer as a convolution + the filter tile: as many filters (a multiple of 8, the MAC kernel's filter granule) as make 2-16 GiB of NTT-form
        weights, by what HBM has left (a tile of 2 filters runs the MAC kernel at a quarter of its rate and costs 250 launches per chunk).
        On the matrix cores (reductions the limb GEMM takes, at least 32 rows per launch): tiles of 64 filters in limb form, built from canonical sub-tiles of 8 filters
        (crc_limb_pack_weights_tile), the layer's input converted to limb form once per launch"""
        g = dict(zd=a["zd"], xd=a["xd"], yd=a["yd"], xs=a["xs"], ys=a["ys"], xf=a["xf"], yf=a["yf"], nf=a["nf"]) if kind == "conv" else \
            dict(zd=a["in_dim"], xd=1, yd=1, xs=1, ys=1, xf=1, yf=1, nf=a["out_dim"])
        T = g["zd"] * g["xf"] * g["yf"]
        g["T"] = T
        g["P"] = ((g["xd"] - g["xf"]) // g["xs"] + 1) * ((g["yd"] - g["yf"]) // g["ys"] + 1)
        g["limb"] = bool(self.limb and B is not None and self.planned_form(kind, a, B) == binding.NTTL)        # (crc_plan_mac: row count and filter count rules included)
        if g["limb"]:
            g["ft"] = min(64, g["nf"]); g["sub"] = min(8, g["ft"])
            return g
        if not hasattr(self, "_stream_tile_bytes"):
            self._stream_tile_bytes = max(2 << 30, min(16 << 30, self.E.mem_info()[0] // 8))
        ft = self._stream_tile_bytes // (T * self.E.k * self.E.n * 8)
        g["ft"] = max(1, min(g["nf"], ft // 8 * 8 if ft >= 8 else ft))
        return g

    def _forward_streamed(self, kind, a, p, cur, B, out):
        """lift + NTT a tile of filters, run the layer on the tile, scatter the tile's output channels into the [B][F][P] tensor"""
        E = self.E
        # (an input that arrives as a group's limb tensor keeps the layer on the matrix cores whatever the size of this particular group)
        g = self._stream_geometry(kind, a, B if p["in_form"] != binding.NTTL else max(B, 32))
        rowb = E.k * E.n * 8; ctb = self.ct_bytes()
        # the layer's input goes to limb form once for all filter tiles (a coefficient-form input -- the reference's layer contract -- is left to every tile's call)
        pre = g["limb"] and p["in_form"] in (NTT, binding.NTTP)
        if pre:
            E.limb_pack_tensor(cur, p["in_form"], B, g["zd"], g["xd"], g["yd"], self.xltile)
        elif g["limb"] and p["in_form"] == binding.NTTL:          # (a group's limb tensor assembled by forward_group)
            pre = True; xl_in = cur
        for f0 in range(0, g["nf"], g["ft"]):
            ft = min(g["ft"], g["nf"] - f0)
            bias = E.p(p["b"]) + f0 * rowb
            if g["limb"]:
                for s0 in range(0, ft, g["sub"]):
                    fs = min(g["sub"], ft - s0)
                    E.plain_to_ntt(E.p(p["plain"]) + (f0 + s0) * g["T"] * E.n * 8, fs * g["T"], self.wtile)
                    E.limb_pack_weights_tile(self.wtile, ft, s0, fs, g["zd"], g["xf"], g["yf"], self.wltile)
                E.conv2d((xl_in if p["in_form"] == binding.NTTL else self.xltile) if pre else cur, self.wltile, bias, B, g["zd"], g["xd"], g["yd"], g["xs"], g["ys"], g["xf"], g["yf"], ft, binding.NTTL if pre else p["in_form"],
                         p["out_form"], self.ytile, self.work, w_form=binding.NTTL)
            else:
                E.plain_to_ntt(E.p(p["plain"]) + f0 * g["T"] * E.n * 8, ft * g["T"], self.wtile)
                E.conv2d(cur, self.wtile, bias, B, g["zd"], g["xd"], g["yd"], g["xs"], g["ys"], g["xf"], g["yf"], ft, p["in_form"], p["out_form"], self.ytile, self.work)
            for b in range(B):
                E.L.crc_memcpy_d2d(E.c, E.p(out) + (b * g["nf"] + f0) * g["P"] * ctb, E.p(self.ytile) + b * ft * g["P"] * ctb, ft * g["P"] * ctb, E.stream)

    def _slots(self):
        """ping-pong slot of every layer's output (in-place layers keep their input's slot); -1 = caller's input"""
        slots, cur = [], -1
        for (kind, *_rest) in self.plan:
            if kind == "bn" and cur >= 0:
                slots.append(cur)
            else:
                cur = 0 if cur != 0 else 1
                slots.append(cur)
        return slots

    def prepare(self, B, limb=True, tail_group=1):
        """allocate the two ping-pong activation buffers and the work space for chunks of B images; put the MAC operands into their kernel's operand form
        (limb=False keeps every layer on the vector-ALU kernel: needed while fuse() is still to come)"""
        for pl_ in self.plan:                               # (a previous prepare() with two-level chunking re-typed the first dense layer's input)
            if "in_form_chunked" in pl_[3]:
                pl_[3]["in_form"] = pl_[3].pop("in_form_chunked")
        self.B = B
        self.G = max(1, int(tail_group))
        # two-level chunking: the layers in front of the first dense layer run on chunks of B images (their activations bound the chunk), the dense layers on
        # G chunks at once: a dense layer streams all of its weights per launch (PlainModelWoPad's fc3 at n = 16384: 177 GiB), so its time per image falls with the
        # number of rows = 2 x images the stream is used for -- and a 64-row matrix-core tile wants 32 images
        self.split = next((i for i, pl_ in enumerate(self.plan) if pl_[0] == "fc"), len(self.plan)) if self.G > 1 else len(self.plan)
        if self.split == 0 or self.split == len(self.plan):
            self.G = 1; self.split = len(self.plan)
        if self.materialize:
            self._build_tilewise()
            if limb:
                # (the split is known first: every layer is planned -- crc_plan_mac's rows-per-launch rule -- for the images IT will be launched on: B in front of the
                # split, B G behind it; a degenerate split runs everything on B)
                self._limb_operands(B, B * self.G, self.split)
            self._pack_operands()
        acts = self.activation_cts()
        self.slots = self._slots()
        size = [1, 1]
        for i, sl in enumerate(self.slots):
            if i >= self.split:
                continue
            size[sl] = max(size[sl], acts[i + 1] * B * self.ct_bytes())
            of = self.plan[i][3].get("out_form")
            if of == binding.NTTLC:          # a limb tensor pads the channels to 32 (7 bytes per residue instead of 8)
                nf, xo, yo = self.plan[i][5]
                size[sl] = max(size[sl], self.E.limb_tensor_bytes(B, nf, xo, yo))
            elif of == binding.NTTL:         # a dense consumer's tensor: all of the layer's outputs as channels of one position, rounded up to 32 (larger than the
                size[sl] = max(size[sl], self.E.limb_tensor_bytes(B, acts[i + 1], 1, 1))      # ciphertexts themselves below 217 channels)
        self.buf = [self.alloc(size[0]), self.alloc(size[1])]
        self.act_bytes = size[0] + size[1]
        self.tail_in = self.tail_buf = None
        self.tail_limb = None
        if self.G > 1:
            Bt = B * self.G
            self.tail_in_img = acts[self.split] * self.ct_bytes()                 # bytes per image of the tensor handed to the first dense layer
            tk, _, ta, t0 = self.plan[self.split][:4]
            on_cores = t0.get("w_form") == binding.NTTL or (t0.get("streamed") and self._stream_geometry(tk, ta, Bt)["limb"])
            if on_cores and t0["in_form"] in (NTT, binding.NTTP):
                # the first dense layer runs on the matrix cores: every chunk's tensor goes straight into ITS limb tensor (crc_limb_pack_tensor_at), 7/8 of the
                # ciphertext bytes and no second conversion buffer inside the layer call (17.6 GiB for PlainModelWoPad's fc3 at 24 images)
                self.tail_limb = dict(form=t0["in_form"], ch=acts[self.split])
                t0["in_form_chunked"] = t0["in_form"]; t0["in_form"] = binding.NTTL
                self.tail_in = self.alloc(self.E.limb_tensor_bytes(Bt, acts[self.split], 1, 1))
            else:
                self.tail_in = self.alloc(Bt * self.tail_in_img)
            tsize = [1, 1]
            for i in range(self.split, len(self.plan)):
                sl = self.slots[i]
                tsize[sl] = max(tsize[sl], acts[i + 1] * Bt * self.ct_bytes())
                if self.plan[i][3].get("out_form") == binding.NTTL:
                    tsize[sl] = max(tsize[sl], self.E.limb_tensor_bytes(Bt, acts[i + 1], 1, 1))
            self.tail_buf = [self.alloc(tsize[0]), self.alloc(tsize[1])]
            self.act_bytes += (self.E.limb_tensor_bytes(Bt, acts[self.split], 1, 1) if self.tail_limb else Bt * self.tail_in_img) + tsize[0] + tsize[1]
        _, work = self.scratch_bytes(B)
        self.work = self.alloc(max(work, 256))
        self.work_bytes = max(work, 256)
        self.wtile = self.ytile = self.wltile = self.xltile = None
        for li, (kind, name, a, p, ishape, oshape) in enumerate(self.plan):
            if p.get("streamed"):
                Bl = B * self.G if li >= self.split else B
                g = self._stream_geometry(kind, a, Bl)
                wt, yt = (g["sub"] if g["limb"] else g["ft"]) * g["T"] * self.E.k * self.E.n * 8, Bl * g["ft"] * g["P"] * self.ct_bytes()
                if self.wtile is None or self._wtile_bytes < wt:
                    self.wtile, self._wtile_bytes = self.alloc(wt), wt
                if self.ytile is None or self._ytile_bytes < yt:
                    self.ytile, self._ytile_bytes = self.alloc(yt), yt
                if g["limb"]:                 # (one set of tile buffers serves every streamed layer: the largest of each)
                    wl = self.E.limb_weights_bytes(g["ft"], g["zd"], g["xf"], g["yf"])
                    xl = self.E.limb_tensor_bytes(Bl, g["zd"], g["xd"], g["yd"]) if p["in_form"] != binding.NTTL else 8         # (a group's limb tensor needs no copy)
                    if self.wltile is None or self._wltile_bytes < wl:
                        self.wltile, self._wltile_bytes = self.alloc(wl), wl
                    if self.xltile is None or self._xltile_bytes < xl:
                        self.xltile, self._xltile_bytes = self.alloc(xl), xl
                p["stream_kernel"] = "mfma_mac2w_kernel on 64-filter limb tiles" if g["limb"] else "mac3_kernel"

    # ---- forward over one chunk of B images; d_x: [B][1][28][28] cts in coefficient form.  Returns device ptr of the
    # [B][10] output cts (coefficient form).  `timer(i, name)` (optional) is called around every layer.
    def forward(self, d_x, B, timer=None):
        assert self.G == 1, "prepared for two-level chunking: use forward_group"
        return self._run(0, len(self.plan), d_x, B, self.buf, timer)

    def forward_group(self, d_xs, B, timer=None):
        """two-level chunking (prepare(..., tail_group=G)): d_xs = up to G chunks of B images each; the layers in front of the first dense layer run chunk by chunk,
        the dense layers once on all of them.  Returns the device pointer of the [len(d_xs) * B][10] output ciphertexts."""
        E = self.E
        if self.G == 1:
            assert len(d_xs) == 1
            return self._run(0, len(self.plan), d_xs[0], B, self.buf, timer)
        assert 1 <= len(d_xs) <= self.G
        Bt = len(d_xs) * B
        for c, d_x in enumerate(d_xs):
            if self.tail_limb:
                cur = self._run(0, self.split, d_x, B, self.buf, timer)
                E.limb_pack_tensor_at(cur, self.tail_limb["form"], B, self.tail_limb["ch"], 1, 1, self.tail_in, Bt, c * B)
            else:
                self._run(0, self.split, d_x, B, self.buf, timer, last_out=E.p(self.tail_in) + c * B * self.tail_in_img)
        return self._run(self.split, len(self.plan), self.tail_in, Bt, self.tail_buf, timer)

    def _run(self, lo, hi, d_x, B, bufs, timer=None, last_out=None):
        E = self.E
        cur = d_x
        for i in range(lo, hi):
            kind, name, a, p, ishape, oshape = self.plan[i]
            out = bufs[self.slots[i]]
            if last_out is not None and i == hi - 1 and kind != "bn":
                out = last_out                   # the last layer in front of the dense layers writes straight into the group's staging tensor
            if timer:
                timer(i, name, kind, 0)
            if p.get("streamed"):
                self._forward_streamed(kind, a, p, cur, B, out)
                cur = out
            elif kind == "conv":
                E.conv2d(cur, p["w"], p["b"], B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"], p["in_form"], p["out_form"], out, self.work,
                         w_form=p.get("w_form", NTT))
                cur = out
            elif kind == "fc":
                E.dense(cur, p["w"], p["b"], B, a["in_dim"], a["out_dim"], p["in_form"], p["out_form"], out, self.work, w_form=p.get("w_form", NTT))
                cur = out
            elif kind in ("pool", "avgpool"):
                E.pool(cur, B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], p["div"], p["form"], out)
                cur = out
            elif kind == "bn":
                if E.p(cur) != E.p(out):     # first layer: never modify the caller's input in place
                    E.L.crc_memcpy_d2d(E.c, E.p(out), E.p(cur), B * int(np.prod(ishape)) * self.ct_bytes(), E.stream)
                    cur = out
                E.batchnorm(cur, B, ishape[0], ishape[1], ishape[2], p["mean"], p["invstd"], p["form"])
            elif kind == "square":
                E.square_relin(cur, B * int(np.prod(ishape)), self.d_evk, out, self.work, self.dbc, p["in_form"], p["out_form"])
                cur = out
            elif kind == "squarepool":
                E.square_pool_relin(cur, B, a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], self.d_evk, out, self.work, self.dbc, p["in_form"], p["out_form"],
                                    d_div=p["div"])          # (an average pooling's divisor multiplies the pooled tensor as it leaves the key switch)
                cur = out
            if timer:
                timer(i, name, kind, 1)
        if last_out is not None and E.p(cur) != last_out:        # (an in-place last layer: copy its tensor over)
            E.L.crc_memcpy_d2d(E.c, last_out, E.p(cur), B * int(np.prod(self.plan[hi - 1][5])) * self.ct_bytes(), E.stream)
        return cur
