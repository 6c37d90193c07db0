"""The Python twin of the measured path: the same workload through crcnn_amd/netrun.py (ctypes over the C ABI).  bench.py uses it (a) for N > 1 -- one rank per
GPU under torch.distributed, RCCL weight broadcast through the engine's C ABI -- and (b) with --python-twin as a cross-check of the C++ host classes' figure.
Also home of the "reference layer structure" passes (every CrCNN layer as its own kernel sequence: the reference's T_LAYER_i columns, mainparams.cpp:81)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from .configs import CONFIGS, ENC_SEED, EVK_SEED, GOLDEN_FOR, HBM_PEAK_GBS, INT8_PEAK_TOPS, KEY_SEED, golden_check
from .cpu import cpu_baseline, cpu_baseline_reference, host_cores
from .geometry import limb_exec_over_useful
from .plain import plain_forward


def run_config(args, D_, cfg_name, steps, warmup, batch=None, chunk=None, full=True):
    """one workload: setup (keys, encrypted inputs, encoded weights + broadcast), `warmup` + `steps` timed passes, verification.
    full=False: the secondary workload (no reference-layer-structure pass, no CPU baseline).  Returns (result dict on rank 0, all checks ok)"""
    import torch
    import crcnn_amd as ca
    from crcnn_amd.netrun import Network, TOPOLOGIES, layer_macs
    rank, world, dev = D_.rank, D_.world, D_.dev

    cfg = dict(CONFIGS[cfg_name])
    if args.t_bits and full:
        cfg["t"] = 1 << args.t_bits
    B = batch or cfg["batch"]; C = min(chunk or cfg["chunk"], B)
    G = max(1, min(int(args.tail or cfg.get("tail", 1)), B // C))       # chunks per run of the dense layers (two-level chunking)
    q = cfg.get("q") or ca.default_coeff_modulus_128(cfg["n"])[:cfg["k"]]
    E = ca.Engine(cfg["n"], q, cfg["t"], device=D_.local)
    E.stream = torch.cuda.current_stream().cuda_stream or None
    keep = []

    def alloc(nbytes):
        if os.environ.get("CRC_BENCH_ALLOC_LOG") and nbytes > (1 << 30):          # (debugging aid: where the HBM goes)
            import traceback
            fr = traceback.extract_stack(limit=3)[0]
            sys.stderr.write(f"alloc {nbytes / 2**30:8.2f} GiB  {os.path.basename(fr.filename)}:{fr.lineno} {fr.name}   (torch allocated {torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB)\n")
        t = torch.empty((int(nbytes) + 7) // 8, dtype=torch.int64, device=dev); keep.append(t); return t

    model = cfg["model"]
    h5 = os.path.join(ROOT, "tests", "golden", "models", model + ".h5")
    W = {nm: ca.h5_read(h5, nm) for nm in ca.h5_list(h5) if not nm.endswith("num_batches_tracked")}

    # ---- keys + encrypted inputs (client side, untimed): `distinct` synthetic images encrypted on the host, tiled on device.
    # Seeded (deterministic, NOT secure) on purpose: image 0 of rank 0 is the input of the reference-made golden (GOLDEN_FOR)
    t_setup = time.time()
    sk, pk = E.keygen(KEY_SEED)
    needs_evk = any(k_ == "square" for k_, _, _ in TOPOLOGIES[model])
    d_evk = evk = None
    if needs_evk:
        evk = E.gen_evk(EVK_SEED, sk)
        d_evk = alloc(evk.nbytes); d_evk.copy_(torch.from_numpy(evk.view(np.int64)))
    from crcnn_amd.synth import normalize, synth_image
    D = max(1, min(args.distinct or 4, B))
    imgs = [normalize(synth_image(rank * 100003 + i)) for i in range(D)]
    ctw = 2 * E.k * E.n
    src = torch.empty((D, 784 * ctw), dtype=torch.int64, device=dev)
    x0_sha = None
    for i, im in enumerate(imgs):
        pl, _ = E.encode(im.reshape(-1))
        ct = E.encrypt(pk, pl, ENC_SEED + 1000 * i)
        if i == 0:
            import hashlib
            x0_sha = hashlib.sha256(np.ascontiguousarray(ct).tobytes()).hexdigest()
        src[i].copy_(torch.from_numpy(ct.reshape(-1).view(np.int64)))
    # the batch is `D` distinct encrypted images tiled B/D times.  It is materialised in HBM when it fits beside the weights
    # (Tiny: 98 GiB); for the bigger rings (1024 x 784 cts is 294 GiB at n=8192) a window of whole chunks is kept instead and
    # chunk c reads window position c mod window -- the same tiling, the same bytes per image
    img_bytes = 784 * ctw * 8
    free_b, total_b = torch.cuda.mem_get_info(dev)
    est_weights = sum((a_.get("nf", 0) * a_.get("zd", 0) * a_.get("xf", 0) * a_.get("yf", 0) + a_.get("in_dim", 0) * a_.get("out_dim", 0)) for _, _, a_ in TOPOLOGIES[model]) * E.k * E.n * 8
    budget = max(img_bytes * C, int(0.45 * (free_b - 2.3 * est_weights)))          # (2.3: the limb copy of the weights is built beside the canonical one)
    step_w = C * D // np.gcd(C, D)                      # window must be a multiple of the chunk and of the tiling period
    window = min(B, max(step_w, (budget // img_bytes) // step_w * step_w)) if budget // img_bytes < B else B
    x_all = alloc(window * img_bytes).view(window, 784 * ctw)
    for b0 in range(0, window, D):
        nb = min(D, window - b0); x_all[b0:b0 + nb].copy_(src[:nb])
    del src

    # ---- encoded weights: rank 0 encodes + NTTs, RCCL broadcast to the others, every rank checksums what it holds
    net = Network(E, model, weights=W, alloc=alloc, resident=(args.mode == "resident"), d_evk=d_evk, materialize=(rank == 0), fuse_pool=False)

    def release(buf):
        keep[:] = [k_ for k_ in keep if k_ is not buf]
    net.release = release
    torch.cuda.synchronize()
    bcast = None
    if world > 1:
        bufs = list(net.param_bufs) + ([(d_evk, d_evk.numel() * 8)] if d_evk is not None else [])
        nbytes = sum(n_ for _, n_ in bufs)
        comm, comm_err = D_.make_comm(E)
        D_.barrier(); t0 = time.time()
        if comm is not None:
            for buf, n_ in bufs:
                E.broadcast_weights(comm, buf, (n_ + 7) // 8 * 8, root=0)          # crc_broadcast_weights: ncclBroadcast over xGMI, <= 1 GiB pieces
            via = "crc_broadcast_weights (RCCL through the engine's C ABI)"
        else:
            from crcnn_amd import shard
            shard.broadcast_buffers([buf for buf, _ in bufs], src=0, chunk_bytes=1 << 30)
            via = f"torch.distributed ({D_.backend})" + (" -- rehearsal backend" if D_.backend != "nccl" else f" -- FALLBACK, crc_comm_create failed: {comm_err}")
        torch.cuda.synchronize(); D_.barrier(); bcast_s = time.time() - t0
        cs = [0, 0]
        for buf, n_ in bufs:
            x_, s_ = E.checksum64(buf, n_ // 8 * 8)
            cs[0] ^= x_; cs[1] = (cs[1] * 0x9E3779B97F4A7C15 + s_) & ((1 << 64) - 1)
        if comm is not None:
            allcs = [tuple(int(v) for v in row) for row in E.allgather_u64(comm, cs)]
            E.comm_destroy(comm)
        else:
            allcs = [tuple(c_) for c_ in D_.gather(cs)]
        bcast = dict(seconds=round(bcast_s, 3), bytes=int(nbytes), GBps=round(nbytes / bcast_s / 1e9, 2), via=via,
                     xgmi_link_peak_GBps=153.0, checksums_match=f"{sum(1 for c_ in allcs if c_ == allcs[0])}/{world}", checksum=f"{allcs[0][0]:016x}:{allcs[0][1]:016x}")
    net.materialize = True                       # every rank now holds the encoded parameters (needed by fuse())
    out_all = alloc(B * 10 * ctw * 8).view(B, 10 * ctw)
    # ---- reference layer structure first (every CrCNN layer run as its own kernel sequence, NTT-resident): a short pass
    unfused = None
    want_fuse = args.mode == "resident" and not args.no_fuse
    tilewise = any(pl[3].get("tilewise") for pl in net.plan)       # (their limb weights would have to be built twice, before and after the folding: skipped)
    prod_ref = None
    est_w_gib = est_weights / 2**30
    if want_fuse and full and args.unfused_images > 0 and not tilewise and world == 1 and est_w_gib < 60:      # (a SECOND encoded network is resident during this pass)
        # the reference's OWN layer structure (no folding: every CrCNN layer its own kernel sequence, NTT-resident) on the PRODUCTION kernels (matrix cores): the
        # reference's T_LAYER_0..N columns (mainparams.cpp:81) for this engine.  A second encoded network (the limb conversion drops the canonical weights that fuse()
        # needs), freed again before the main pass
        mark = len(keep)
        net_r = Network(E, model, weights=W, alloc=alloc, resident=True, d_evk=d_evk, materialize=True, fuse_pool=False)
        net_r.release = release
        Cu = min(C, 16)                                # (two encoded networks are resident during this pass: a smaller chunk of unfused activations)
        nu = min(B, max(Cu, args.unfused_images // Cu * Cu))
        net_r.prepare(Cu, limb=True)
        net_r.forward(x_all[0], 1); torch.cuda.synchronize()
        lev = []

        def timer_r(i, name, kind, phase):
            e = torch.cuda.Event(enable_timing=True); e.record(); lev.append((i, e))
        t0 = time.perf_counter()
        for c0 in range(0, nu, Cu):
            d_out = net_r.forward(x_all[c0 % window], min(Cu, nu - c0), timer=timer_r)
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), min(Cu, nu - c0) * 10 * ctw * 8, E.stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tl = np.zeros(len(net_r.plan))
        for j in range(0, len(lev), 2):
            tl[lev[j][0]] += lev[j][1].elapsed_time(lev[j + 1][1])
        prod_ref = dict(images=nu, images_per_s=round(nu / dt, 3), ms_per_image=round(dt / nu * 1e3, 3), first_outputs=out_all[:min(D, nu)].clone(),
                        T_LAYER={f"T_LAYER_{i}": round(float(tl[i] / nu), 4) for i in range(len(net_r.plan))},
                        layers=[pl[1] for pl in net_r.plan],
                        mac_kernel_per_layer={pl[1]: {ca.NTTL: "mfma_mac2w_kernel", ca.NTTL1: "mfma_conv1_kernel"}.get(pl[3].get("w_form"), "mac3_kernel") for pl in net_r.plan if pl[0] in ("conv", "fc")},
                        note="ms per image and layer of the UNFUSED network in the column order of the reference's timing rows (mainparams.cpp:81), matrix-core kernels")
        del net_r, lev, d_out
        del keep[mark:]
        import gc
        gc.collect(); torch.cuda.empty_cache()
    if want_fuse and full and args.unfused_images > 0 and not tilewise:
        Cu = min(C, 32)                                # the unfused conv1 output is 18 432 ciphertexts per image (Tiny): a smaller chunk than the main pass
        nu = min(B, max(Cu, args.unfused_images // Cu * Cu))
        net.prepare(Cu, limb=False)                    # (the limb conversion drops the canonical weights fuse() needs; it happens in the final prepare)
        net.forward(x_all[0], 1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c0 in range(0, nu, Cu):
            d_out = net.forward(x_all[c0 % window], min(Cu, nu - c0))
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), min(Cu, nu - c0) * 10 * ctw * 8, E.stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        unfused = dict(images=nu, images_per_s=round(nu / dt, 3), ms_per_image=round(dt / nu * 1e3, 3), first_outputs=out_all[:min(D, nu)].clone(),
                       layers=[pl[1] for pl in net.plan], kernels="vector ALU only (mac3_kernel: the canonical weights are still needed by the folding that follows)")
        for t_ in list(net.buf) + [net.work]:          # give the large unfused activation buffers back before the main pass
            keep[:] = [k_ for k_ in keep if k_ is not t_]
        del net.buf, net.work, t_
        torch.cuda.empty_cache()
    if want_fuse:
        net.fuse()            # fold avg/sum pooling into the preceding convolution where that removes MACs (exact)
    net.prepare(C, tail_group=G)
    G = net.G
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    nl = len(net.plan)
    lay_ev = []
    # groups of up to G full chunks (the dense layers run once per group); a ragged last chunk is a group of its own
    groups, c0 = [], 0
    while c0 < B:
        cb = min(C, B - c0)
        ng = min(G, (B - c0) // C) if cb == C else 1
        groups.append((c0, cb, max(1, ng))); c0 += cb * max(1, ng)

    def step(record):
        for (c0, cb, ng) in groups:
            evs = []

            def timer(i, name, kind, phase):
                if record:
                    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((i, e))
            d_out = net.forward_group([x_all[(c0 + j * cb) % window] for j in range(ng)], cb, timer=timer)
            E.L.crc_memcpy_d2d(E.c, out_all[c0].data_ptr(), E.p(d_out), ng * cb * 10 * ctw * 8, E.stream)
            if record:
                lay_ev.append((cb, ng, evs))

    # untimed module-load pass on a single image (not a step)
    net.forward_group([x_all[0]], 1)
    torch.cuda.synchronize()
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    D_.barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        step(s == steps - 1)
    torch.cuda.synchronize()
    D_.barrier()
    elapsed = D_.max(time.perf_counter() - t0)

    # ---- per-layer times of the last step (HIP events on the launch stream)
    lay_ms = np.zeros(nl); lay_launch = np.zeros(nl); lay_cnt = np.zeros(nl)
    for cb, ng, evs in lay_ev:
        for j in range(0, len(evs), 2):                  # (layer i, start), (layer i, end) -- a head layer once per chunk of the group, a dense layer once per group
            i = evs[j][0]; ms = evs[j][1].elapsed_time(evs[j + 1][1])
            lay_ms[i] += ms
            if cb == C and (i < net.split or ng == G):
                lay_launch[i] += ms; lay_cnt[i] += 1
    ms_per_layer = {net.plan[i][1]: round(float(lay_ms[i] / B), 4) for i in range(nl)}

    # ---- verification outside the timed region: tiled images give identical outputs; decrypted logits match the plain model;
    # the output ciphertexts of image 0 are, bit for bit, the compiled reference's (golden fixture)
    ok_tile = all(bool(torch.equal(out_all[b], out_all[b % D])) for b in range(D, B, max(1, (B - D) // 16)))
    ok_fused = True
    if unfused is not None:     # folding pooling into the convolution must not change a single output bit
        fo = unfused.pop("first_outputs")
        ok_fused = unfused["outputs_identical_to_fused"] = bool(torch.equal(fo, out_all[:fo.shape[0]]))
    if prod_ref is not None:
        fo = prod_ref.pop("first_outputs")
        prod_ref["outputs_identical_to_fused"] = bool(torch.equal(fo, out_all[:fo.shape[0]]))
        ok_fused = ok_fused and prod_ref["outputs_identical_to_fused"]
        if unfused is None:
            unfused = {}
        unfused["production_kernels"] = prod_ref
    outs = out_all[:D].cpu().numpy().view(np.uint64).reshape(D, 10, 2, E.k, E.n)
    import hashlib
    gold_ok, gold_name = golden_check(cfg_name, cfg, q, rank, x0_sha, hashlib.sha256(np.ascontiguousarray(outs[0]).tobytes()).hexdigest())
    # BASELINE configs[0] in full (tests/golden/c1_tiny4096_t32.json: 32 images through the compiled reference): this run's distinct images ARE its first images
    c1_ok = None
    c1_path = os.path.join(ROOT, "tests", "golden", "c1_tiny4096_t32.json")
    if rank == 0 and cfg_name == "tiny4096" and os.path.exists(c1_path):
        c1 = json.load(open(c1_path))
        if (c1["t"], [int(v) for v in c1["q"]], c1["key_seed"], c1["enc_seed_base"], c1["enc_seed_stride"]) == (cfg["t"], [int(v) for v in q], KEY_SEED, ENC_SEED, 1000):
            have = [i for i in range(D) if str(i) in c1["images"]]
            hits = sum(1 for i in have if c1["images"][str(i)]["out_sha256"] == hashlib.sha256(np.ascontiguousarray(outs[i]).tobytes()).hexdigest())
            c1_ok = f"{hits}/{len(have)}"
    preds_ok, budgets, max_err = 0, [], 0.0
    for i in range(D):
        dec = E.decrypt(sk, outs[i])
        logits = np.array([E.decode(dec[j]) for j in range(10)])
        want = plain_forward(model, W, imgs[i])
        budgets.append(E.noise_budget(sk, outs[i][0]))
        max_err = max(max_err, float(np.abs(logits - want).max()))
        preds_ok += int(np.argmax(logits) == np.argmax(want))
    mine_ok = bool(ok_tile and ok_fused and preds_ok == D and gold_ok is not False and (c1_ok is None or c1_ok.split("/")[0] == c1_ok.split("/")[1]))
    ranks_ok = D_.sum(int(mine_ok))               # every rank must have verified its own outputs
    all_ok = ranks_ok == world and (bcast is None or bcast["checksums_match"] == f"{world}/{world}")

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: algorithmic bytes per launch / measured duration
        dom = int(np.argmax(lay_launch))
        kind, name, a, p, ishape, oshape = net.plan[dom]
        in_cts, out_cts = int(np.prod(ishape)), int(np.prod(oshape))
        ct_bytes = 8 * E.k * E.n * 2
        wbytes = 0
        if kind == "conv":
            wbytes = a["nf"] * a["zd"] * a["xf"] * a["yf"] * 8 * E.k * E.n
        elif kind == "fc":
            wbytes = a["in_dim"] * a["out_dim"] * 8 * E.k * E.n
        CL = C * G if dom >= net.split else C           # images per launch of that layer
        alg_bytes = CL * (in_cts + out_cts) * ct_bytes + wbytes
        dur_ms = lay_launch[dom] / max(1, lay_cnt[dom])
        layer_ms = dur_ms
        kernel_note = None
        if p.get("w_form") == ca.NTTL and kind in ("conv", "fc") and dom > 0 and p["in_form"] != ca.NTTL:
            # the layer call is three kernels (tensor -> limb form, mfma_mac_kernel, result -> the next layer's form).  Time the MAC kernel itself: regenerate the
            # layer's input for one chunk with the layers in front of it, convert it once, then launch the layer on the limb tensor (HIP events, same stream)
            try:
                cur = x_all[0]
                for li in range(dom):
                    k_, n_, a_, p_, is_, os_ = net.plan[li]
                    assert k_ == "conv"
                    E.conv2d(cur, p_["w"], p_["b"], C, a_["zd"], a_["xd"], a_["yd"], a_["xs"], a_["ys"], a_["xf"], a_["yf"], a_["nf"], p_["in_form"], p_["out_form"], net.buf[net.slots[li]],
                             net.work, w_form=p_.get("w_form", ca.NTT))
                    cur = net.buf[net.slots[li]]
                gz = (a["zd"], a["xd"], a["yd"], a["xs"], a["ys"], a["xf"], a["yf"], a["nf"]) if kind == "conv" else (a["in_dim"], 1, 1, 1, 1, 1, 1, a["out_dim"])
                xl = alloc(E.limb_tensor_bytes(C, gz[0], gz[1], gz[2]))
                E.limb_pack_tensor(cur, p["in_form"], C, gz[0], gz[1], gz[2], xl)
                outk = net.buf[net.slots[dom]]
                run_k = lambda: E.conv2d(xl, p["w"], p["b"], C, gz[0], gz[1], gz[2], gz[3], gz[4], gz[5], gz[6], gz[7], ca.NTTL, ca.NTTL, outk, net.work, w_form=ca.NTTL)
                run_k(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run_k()
                e1.record(); torch.cuda.synchronize()
                dur_ms = e0.elapsed_time(e1) / 5
                kernel_note = (f"mfma_mac2w_kernel timed on its own (+ the 3 % slotmajor_to_limb conversion behind it): 5 launches on the limb-form input of one chunk; the whole layer call "
                               f"(limb_pack_tensor + mfma_mac2w_kernel + conversion) takes {layer_ms:.2f} ms inside the timed region")
            except Exception as ex:          # keep the layer-level figure
                kernel_note = f"kernel-only timing failed ({type(ex).__name__}); launch_ms is the whole layer call"
        if kernel_note is None and p.get("w_form") == ca.NTTL:
            kernel_note = ("HIP events around the layer call inside the timed region; the input arrives in limb form from the layer in front and the call is " +
                           ("mfma_mac2w_kernel alone (it writes the next dense layer's limb tensor itself)" if p["out_form"] == ca.NTTL else "mfma_mac2w_kernel + the conversion of its slot-major result (3-9 % of the call)"))
        achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        macs_launch = layer_macs(kind, a) * CL
        # HBM traffic of that launch: rocprofv3 PMC passes (FETCH_SIZE corrected x2 for gfx950, WRITE_SIZE) collected OFFLINE with
        # tools/bench_mac.py and committed under profiles/ -- bench.py cannot run the profiler on itself, so this is not measured in this run
        traffic, traffic_source = None, None
        kname = "mfma_mac2w_kernel" if p.get("w_form") == ca.NTTL else "mfma_conv1_kernel" if p.get("w_form") == ca.NTTL1 else "mac3_kernel"
        kernel_label = f"{kname} ({name}, {CL} images/launch)" if kind in ("conv", "fc") else f"{kind} ({name})"
        for pf in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", pf))).get(cfg_name)
                if pm and pm.get("per_ciphertext") and kind == "square":      # the Square + relinearise sequence: PMC bytes per ciphertext x the launch's ciphertexts
                    traffic = int(pm["traffic_bytes_per_ciphertext"] * CL * in_cts)
                    traffic_source = f"profiles/{pf} ({pm['kernel']}), offline rocprofv3 --pmc passes of tools/bench_square.py on the same ring (not measured in this run)"
                    break
                if pm and pm.get("pooled") and kind == "squarepool":          # one key switch per pooled ciphertext: PMC bytes per SQUARED ciphertext
                    traffic = int(pm["pooled"]["traffic_bytes_per_ciphertext"] * CL * in_cts)
                    traffic_source = f"profiles/{pf} ({pm['pooled']['kernel']}), offline rocprofv3 --pmc passes of tools/bench_square_pool.py on the same ring (not measured in this run)"
                    break
                if pm and pm["kernel"] == kernel_label:
                    traffic = int(pm["traffic_bytes"]); traffic_source = f"profiles/{pf}, offline rocprofv3 --pmc passes of the same launch (not measured in this run)"
                    break
            except Exception:
                pass
        modmul_s = macs_launch * 2 * E.k * E.n / (dur_ms * 1e-3) if dur_ms > 0 and macs_launch else None
        if kname == "mfma_mac2w_kernel" and modmul_s:
            # the matrix-core kernel is bound by the int8 MFMA rate, not by HBM: 49 limb products (98 int8 operations) per modular multiply-add, against the dense
            # int8 peak (2x the bf16 rate per clock: MI355X_MICROARCH.md, matrix cores).  The HBM view of the same launch stays beside it
            tops = modmul_s * 98 / 1e12
            # executed = what the matrix cores really multiply (limb_exec_over_useful); useful = the layer's own multiply-adds
            exec_over_useful = limb_exec_over_useful(kind, a, CL, out_cts)
            roofline = dict(bound="mfma", achieved=round(tops * exec_over_useful, 1), peak=INT8_PEAK_TOPS, unit="TOP/s (int8)", frac=round(tops * exec_over_useful / INT8_PEAK_TOPS, 5),
                            useful_achieved=round(tops, 1), useful_frac=round(tops / INT8_PEAK_TOPS, 5), traffic=traffic,
                            ops="int8 multiply and add, 98 per modular multiply-add (7 x 7 balanced base-256 limb products); useful = the layer's ct x pt multiply-adds x 2 polys x k n, "
                                "executed = the same with rows / channels / filters padded to the kernel's tiles",
                            hbm_achieved_GBps=round(achieved, 2), hbm_frac=round(achieved / HBM_PEAK_GBS, 5))
        else:
            roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic)
        roofline.update(traffic_source=traffic_source, kernel=kernel_label, kernel_timing=kernel_note, launch_ms=round(float(dur_ms), 3), layer_call_ms=round(float(layer_ms), 3),
                        algorithmic_bytes_per_launch=int(alg_bytes), modmul_per_s=round(modmul_s, 1) if modmul_s else None)
        cpu = None
        if args.cpu_seconds > 0 and world == 1:      # (the CPU leg runs at N = 1 only: at N > 1 the host cores are busy driving N ranks)
            x0 = x_all[0].cpu().numpy().view(np.uint64).reshape(1, 28, 28, 2, E.k, E.n)
            cpu = cpu_baseline_reference(cfg, q, W, x0, host_cores(), evk=evk) or cpu_baseline(cfg, q, W, x0, args.cpu_seconds)
            cpu["value"] = round(cpu["value"], 6); cpu["mac_per_s"] = round(cpu["mac_per_s"], 1)
            try:            # configs[0] measured in full in the build container (not extrapolated): oracle/make_c1.py
                c1f = json.load(open(c1_path))
                if cfg_name == "tiny4096":
                    cpu["c1_in_full"] = dict(images=len(c1f["images"]), images_per_s=c1f.get("images_per_s_adjusted", c1f["images_per_s"]),
                                             total_wall_s=c1f.get("total_wall_s_adjusted", c1f["total_wall_s"]), threads=c1f["ref_threads"], note=c1f.get("note"),
                                             where="build container (8 cores), the compiled reference on 32 images: tests/golden/c1_tiny4096_t32.json")
            except Exception:
                pass
        value = B * world * steps / elapsed
        result = {
            "metric": "encrypted images/sec", "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": f"synthetic ({D} distinct MNIST-like encrypted images per GPU tiled to the batch" + ("" if window == B else f", resident as a {window}-image window") + f"; trained weights from {model}.h5)",
            "config": {"workload": f"{model}.h5 n={cfg['n']} k={cfg['k']} t=2^{cfg['t'].bit_length() - 1} batch={B}/GPU chunk={C}" + (f" (dense layers: {C * G})" if G > 1 else "") + f" ({cfg_name}, BASELINE configs)",
                       "mode": args.mode + ("+conv/pool folding" if want_fuse else ""), "parallelism": f"image-sharded x{world}, RCCL weight broadcast"},
            "ms_per_layer": ms_per_layer,
            "mac_kernel_per_layer": {pl[1]: ("mfma_mac2w_kernel (int8 limb GEMM, CRC_NTTL)" if pl[3].get("w_form") == ca.NTTL else
                                             "mfma_conv1_kernel (one-channel convolution on the matrix cores, CRC_NTTL1)" if pl[3].get("w_form") == ca.NTTL1 else
                                             (pl[3]["stream_kernel"] + ", streamed weights (coefficient-form plaintexts lifted + transformed a filter tile at a time)" if pl[3].get("streamed") else
                                              "mac3_kernel (v_mad_u64_u32, CRC_NTTP)") +
                                             (f" [{pl[3]['limb_skipped']}]" if pl[3].get("limb_skipped") else "")) for pl in net.plan if pl[0] in ("conv", "fc")},
            "mfma_useful_frac_per_layer": {pl[1]: round(1.0 / limb_exec_over_useful(pl[0], pl[2], C * G if li >= net.split else C, int(np.prod(pl[5]))), 4)
                                           for li, pl in enumerate(net.plan) if pl[0] in ("conv", "fc") and pl[3].get("w_form") == ca.NTTL},
            "reference_layer_structure": unfused, "roofline": roofline, "cpu_baseline": cpu,
            "check": {"tiled_outputs_identical": bool(ok_tile), "predictions_match_plain_model": f"{preds_ok}/{D}", "max_logit_abs_err": round(max_err, 6),
                      "noise_budget_bits": budgets, "ranks_verified": f"{ranks_ok}/{world}", "golden_match": gold_ok, "golden": gold_name, "c1_images_match_reference": c1_ok, "all_ok": bool(all_ok)},
            "hbm_plan": dict(total_bytes=int(total_b), free_after_setup=int(torch.cuda.mem_get_info(dev)[0]), parameters=int(net.weight_bytes), input_window=int(window * img_bytes),
                             activation_and_work_buffers=int(sum(t_.numel() * 8 for t_ in list(getattr(net, "buf", [])) + [net.work] if hasattr(t_, "numel"))),
                             outputs=int(B * 10 * ctw * 8), evaluation_keys=int(d_evk.numel() * 8) if d_evk is not None else 0,
                             note="bytes on this rank; parameters = encoded weights in their kernels' operand forms (a tile-wise layer's limb tensor is built on every rank from "
                                  "the float weights and is not part of the broadcast)"),
            "setup_s": round(setup_s, 1), "weight_broadcast": bcast, "weight_broadcast_s": bcast["seconds"] if bcast else 0.0, "weight_bytes": int(net.weight_bytes),
        }
    # give everything back before a second workload
    del net, x_all, out_all, outs, keep[:]
    E.sync(); E.close()
    torch.cuda.empty_cache()
    return result, bool(all_ok)
