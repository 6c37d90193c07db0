#!/usr/bin/env python3
"""bench.py -- encrypted images/sec of the CrCNN evaluation path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

One "step" = one pass of the hot path (Network::forward: conv/pool/[bn/square]/dense over every image) over one batch of `--batch` synthetic encrypted
MNIST-like images per GPU.  Rank 0 prints ONE JSON line.

The measured path (N = 1).  north_star: "C++ host code calls hand-written HIP kernels through a thin C-ABI" -- so `value` is what crcnn_amd/lib/bench_host measures:
the C++ host classes (crcnn_amd/host: CnnBuilder builds the network from the HDF5 model, Network::fuse folds it, Network::forward runs it; the reference's class names
and signatures) over include/crcnn_hip.h.  This script is the harness around it: the client side (keys, encrypted inputs: tools/benchkit/client.py), the child process,
the verification of the ciphertexts that come back (the compiled reference's goldens, decrypted logits), the roofline arithmetic and the CPU baseline.
  --python-twin  additionally times the same workload through the Python twin (crcnn_amd/netrun.py, ctypes over the same C ABI) as a cross-check
N > 1: the SAME measured path, one bench_host process per GPU.  Every rank of this script (started by torch.distributed.run, or by the script itself when it is
given --gpus N without a launcher) prepares its own encrypted images on the host cores and starts its bench_host child -- the Python ranks never touch a GPU.  Rank 0's
child makes the RCCL rendezvous id (crc_comm_unique_id) and leaves it in a file, every child joins (crc_comm_create), the encoded model goes out once with
Network::broadcastParameters, and the children bracket their timed steps with an all-gather + stream synchronisation and gather the elapsed times: `value` = all ranks'
images / the slowest rank's time.  The Python ranks only rendezvous over gloo (CPU) to agree on the file name and to count the ranks whose outputs verified.
  --python-twin at N > 1: the round-4 path (every rank runs netrun.py under torch.distributed) instead
  --stream-inputs ciphertext|plaintext|both: besides the resident measurement, passes in which every launch's images are uploaded over PCIe while the previous
                launch is evaluated (two device buffers, a copy stream beside the compute stream); reported under "streamed"

Besides the contract fields the line carries
  roofline      the dominant kernel against the roof that bounds it: int8 matrix-core TOP/s for the limb GEMM (frac = executed, useful_frac = without padding; the HBM view
                of the same launch beside it), HBM GB/s (algorithmic bytes / measured launch duration vs the 8 TB/s peak) for everything else.  Launch durations are HIP
                events on the launch stream inside the timed region (Network::time_with_events)
  cpu_baseline  the compiled reference (oracle/_ref/ref_harness; the CPU oracle where that binary is absent) on the host cores, on a bounded sample
  ms_per_layer  per-image milliseconds per (fused) layer; hbm_plan: where the rank's HBM goes
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from benchkit.configs import CONFIGS, HBM_PEAK_GBS, INT8_PEAK_TOPS, KEY_SEED  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--config", default="tiny4096", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="encrypted images per GPU per step (default: the config's)")
    ap.add_argument("--chunk", type=int, default=None, help="images processed per layer launch")
    ap.add_argument("--tail", type=int, default=None, help="chunks per launch of the dense layers (two-level chunking; default: the config's)")
    ap.add_argument("--distinct", type=int, default=None, help="distinct encrypted images, tiled to the batch (default: the config's -- 32 for tiny4096 = BASELINE configs[0]'s "
                    "images, every one checked against the compiled reference; 32 for approx8192; 12 for wopad16384, whose images are 784 MiB each)")
    ap.add_argument("--no-fuse", action="store_true", help="do not fold pooling / batch-norm layers (Network::fuse)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--also", default="auto", help="further workloads measured in the same invocation and reported under \"also\" (auto: AUTO_ALSO below beside the tiny4096 "
                    "headline -- BASELINE configs[2] and configs[4]'s per-GPU share, k = 4 at n = 8192, the no-MFMA pass, the reference's two published configurations; none: skip)")
    ap.add_argument("--also-steps", type=int, default=2)
    ap.add_argument("--also-batch", type=int, default=None)
    ap.add_argument("--python-twin", action="store_true", help="N = 1: also time the workload through the Python twin (netrun.py) and report it beside the C++ figure")
    # options of the Python twin (N > 1, --python-twin)
    ap.add_argument("--mode", default="resident", choices=["resident", "layerwise"])
    ap.add_argument("--unfused-images", type=int, default=0, help="twin only: images of an extra pass with every reference layer run separately (the reference's T_LAYER columns)")
    ap.add_argument("--t-bits", type=int, default=None, help="twin only: override the plain modulus t = 2^bits")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, rendezvous (gloo, no GPU call) and report: CPU test of the self-launch path")
    ap.add_argument("--stream-inputs", default="auto", choices=["auto", "none", "ciphertext", "plaintext", "both"],
                    help="also measure with the input launches streamed over PCIe (double-buffered upload beside the kernels): 784 ciphertexts per image, or the 784 pixel "
                         "plaintexts per image + encryption on the device (auto: both for the tiny4096 headline and ciphertext for approx8192 of the default invocation, "
                         "none otherwise)")
    ap.add_argument("--stream-steps", type=int, default=2, help="passes over the batch of the streamed measurement")
    ap.add_argument("--latency", default="auto", choices=["auto", "on", "off"], help="single-image latency lines (batch 1, a synchronisation per image) for the headline and the "
                    "BASELINE configurations of the default invocation (auto: with the default invocation at N = 1)")
    ap.add_argument("--weights-via", default="broadcast", choices=["broadcast", "floats"], help="N > 1: how every rank gets the encoded model -- broadcast: rank 0 encodes + "
                    "transforms, Network::broadcastParameters sends the residues over RCCL; floats: every rank encodes + transforms from the model file itself (SURVEY 8e's "
                    "alternative), only the evaluation keys and the checksums travel")
    ap.add_argument("--mnist-dir", default=os.environ.get("CRC_MNIST_DIR", ""), help="directory with t10k-images-idx3-ubyte (and t10k-labels-idx1-ubyte): when present the "
                    "distinct images are real MNIST test images and the line reports agreement with the float model's predictions (utils.cpp:20-53)")
    return ap.parse_args()


def sizes(args, cfg, batch=None):
    B = batch or cfg["batch"]
    C = min(args.chunk or cfg["chunk"], B)
    G = max(1, min(int(args.tail or cfg.get("tail", 1)), B // C))
    return B, C, G


_INPUTS = {}        # the client side of the configuration measured last: keys, images, the encrypted inputs file (one configuration's inputs in /dev/shm at a time)


def drop_inputs():
    # (CRC_BENCH_KEEP: the command lines stay; the encrypted inputs -- gigabytes per configuration -- stay only for the configurations CRC_BENCH_KEEP_CONFIGS names)
    keep_dir = os.environ.get("CRC_BENCH_KEEP")
    keep_cfgs = [c_ for c_ in os.environ.get("CRC_BENCH_KEEP_CONFIGS", "").split(",") if c_]
    for c in _INPUTS.values():
        if not keep_dir:
            shutil.rmtree(c["work"], ignore_errors=True)
        elif keep_cfgs and os.path.basename(c["work"]) not in keep_cfgs:
            for f in ("inputs.u64", "plain_inputs.u64", "outputs.u64", "outputs.u64.streamed"):
                try:
                    os.remove(os.path.join(c["work"], f))
                except OSError:
                    pass
    _INPUTS.clear()


def client_inputs(args, cfg_name, cfg, q, D, rank, cores):
    """keys + the D distinct encrypted images of (model, parameters, rank): made once and shared by the measurements of that parameter set that follow each other
    (throughput, single-image latency, the no-matrix-core pass) -- the inputs are the same ciphertexts, and encrypting them is the longest untimed step"""
    from benchkit.client import Client
    key = (cfg["model"], cfg["n"], tuple(int(v) for v in q), cfg["t"], rank)
    c = _INPUTS.get(key)
    if c and c["D"] >= D and c["mnist_dir"] == args.mnist_dir:
        return c, True
    drop_inputs()
    t_client = time.time()
    client = Client(cfg, q, rank=rank)
    imgs, mnist = client.images_or_mnist(D, args.mnist_dir)
    # (CRC_BENCH_KEEP=<dir>: the encrypted inputs and the bench_host command line stay there afterwards -- tools/measure_round.sh profiles that command with rocprofv3)
    keep_dir = os.environ.get("CRC_BENCH_KEEP")
    if keep_dir:
        work = os.path.join(keep_dir, cfg_name); os.makedirs(work, exist_ok=True)
    else:
        work = tempfile.mkdtemp(prefix="crc_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    x0_sha = client.encrypt_images(imgs, os.path.join(work, "inputs.u64"), threads=cores)
    c = dict(client=client, imgs=imgs, mnist=mnist, work=work, x0_sha=x0_sha, D=D, mnist_dir=args.mnist_dir, client_s=time.time() - t_client)
    _INPUTS[key] = c
    return c, False


def run_host(args, cfg_name, steps, warmup, batch=None, device=0, ranks=None, stream="none", over=None, extra=None, with_cpu=True):
    """one workload through crcnn_amd/lib/bench_host.  Returns (bench line dict, all checks ok).  ranks: benchkit.dist.HostRanks when this process is one of several
    (one bench_host per GPU); only rank 0 gets a line back.  over: chunk / tail / distinct of this run (default: the command line's, then the configuration's);
    extra: further bench_host arguments (sync_each=1 ...)"""
    import crcnn_amd as ca
    from benchkit import cpu, geometry
    cfg = dict(CONFIGS[cfg_name])
    over = over or {}
    B = batch or cfg["batch"]
    C = min(over.get("chunk") or args.chunk or cfg["chunk"], B)
    G = max(1, min(int(over.get("tail") or args.tail or cfg.get("tail", 1)), B // C))
    q = cfg.get("q") or ca.default_coeff_modulus_128(cfg["n"])[:cfg["k"]]
    D = max(1, min(over.get("distinct") or args.distinct or cfg.get("distinct", 4), C * G, B))
    rank, world = (ranks.rank, ranks.world) if ranks else (0, 1)
    cores = cpu.host_cores(ranks.local_world if ranks else 1)
    keep_dir = os.environ.get("CRC_BENCH_KEEP")
    ci, reused = client_inputs(args, cfg_name, cfg, q, D, rank, cores)
    client, imgs, mnist, work, x0_sha = ci["client"], ci["imgs"][:D], ci["mnist"], ci["work"], ci["x0_sha"]
    client.cfg = cfg                                                    # (shared inputs: this run's configuration decides what verify() requires)
    if mnist is not None:
        mnist = {k_: (v[:D] if isinstance(v, list) else v) for k_, v in mnist.items()}
    client_s = 0.0 if reused else ci["client_s"]
    try:
        ctw = 2 * cfg["k"] * cfg["n"]
        x0 = np.fromfile(os.path.join(work, "inputs.u64"), dtype=np.uint64, count=784 * ctw).reshape(1, 28, 28, 2, cfg["k"], cfg["n"])
        exe = os.path.join(ROOT, "crcnn_amd", "lib", "bench_host")
        if not os.path.exists(exe):
            raise SystemExit("bench.py: crcnn_amd/lib/bench_host has not been built (python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        cmd = [exe, f"model={cfg['model']}", "h5=" + os.path.join(ROOT, "tests", "golden", "models", cfg["model"] + ".h5"), f"n={cfg['n']}", f"k={cfg['k']}", f"t={cfg['t']}",
               "q=" + ",".join(str(int(v)) for v in q), "inputs=" + os.path.join(work, "inputs.u64"), f"distinct={D}", f"batch={B}", f"chunk={C}", f"group={G}", f"steps={steps}",
               f"warmup={warmup}", "outputs=" + os.path.join(work, "outputs.u64"), f"fuse={0 if args.no_fuse else 1}", f"key_seed={KEY_SEED}", f"device={device}"]
        if "reenc" in cfg:
            cmd.append(f"reenc={cfg['reenc']}")
        if "matrix_cores" in cfg:
            cmd.append(f"matrix_cores={cfg['matrix_cores']}")
        if ranks and args.weights_via != "broadcast":
            cmd.append(f"weights_via={args.weights_via}")
        cmd += [f"{k_}={v}" for k_, v in (extra or {}).items()]
        if ranks:
            cmd += [f"rank={rank}", f"world={world}", f"local_world={ranks.local_world}", "rendezvous=" + ranks.rendezvous_path(cfg_name)]
        modes = {"none": [], "both": ["ciphertext", "plaintext"]}.get(stream, [stream])
        if modes:
            if "plaintext" in modes:
                client.write_plain_images(imgs, os.path.join(work, "plain_inputs.u64"))
                cmd.append("plain_inputs=" + os.path.join(work, "plain_inputs.u64"))
            cmd += ["stream_inputs=" + ",".join(modes), f"stream_steps={max(1, args.stream_steps)}"]
        if keep_dir:
            open(os.path.join(work, f"cmd_{cfg_name}{'_b1' if B == 1 else ''}.txt"), "w").write(" ".join(cmd) + "\n")
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC (RCCL across processes)
        # A child that dies leaves the others inside a collective nobody will complete (ncclCommInitRank, the broadcast): the rank whose child failed drops a marker
        # beside the rendezvous file, every harness polls for it while its own child runs and ends that child -- the job stops in seconds, not at a timeout.
        marker = ranks.rendezvous_path(cfg_name) + ".failed" if ranks else None
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        t_end, killed = time.time() + 3000, False
        while True:
            try:
                out_s, err_s = proc.communicate(timeout=2.0)
                break
            except subprocess.TimeoutExpired:
                if (marker and os.path.exists(marker)) or time.time() > t_end:
                    proc.kill(); killed = True
        failed = None
        # exit code 4 = the run completed but one of bench_host's own comparisons failed (a streamed or the last timed launch differs from the first): its JSON line is
        # there and names the check -- that is a failed self-check of this rank (reported under "check", exit code 3 of this script), not a crash that stops the job
        if proc.returncode not in (0, 4):
            failed = (f"bench.py: bench_host on rank {rank} was stopped because another rank's failed" if killed and marker and os.path.exists(marker) else
                      f"bench.py: bench_host failed on rank {rank} (exit {proc.returncode}): {err_s[-600:]}")
            if marker and not killed:
                open(marker, "w").write(str(rank))
        r = outs = outs_pt = None
        if failed is None:
            try:                                                    # (a short outputs file or a broken line must not leave the other ranks waiting in ranks.any)
                r = json.loads([ln for ln in out_s.splitlines() if ln.startswith("{")][-1])
                outs = np.fromfile(os.path.join(work, "outputs.u64"), dtype=np.uint64).reshape(D, 10, 2, cfg["k"], cfg["n"])
                if "plaintext" in modes:
                    outs_pt = np.fromfile(os.path.join(work, "outputs.u64.streamed"), dtype=np.uint64).reshape(D, 10, 2, cfg["k"], cfg["n"])
            except Exception as e:
                failed = f"bench.py: rank {rank}: cannot read bench_host's results ({type(e).__name__}: {e}); stderr: {err_s[-400:]}"
        if ranks and ranks.any(failed is not None):                # every rank learns of a failure anywhere: nobody waits for a line that will not come
            raise SystemExit(failed or f"bench.py: bench_host failed on another rank (this is rank {rank})")
        if failed:
            raise SystemExit(failed)
    except BaseException:
        drop_inputs()
        raise
    check, ok = client.verify(cfg_name, imgs, outs, x0_sha, golden=mnist is None)
    if mnist is not None:               # real test images: agreement of the decrypted predictions with the labels and with the float model's predictions the reference ships
        preds = client.last_predictions
        mnist = dict(mnist, encrypted_predictions=preds)
        if "labels" in mnist:
            mnist["agree_with_labels"] = f"{sum(int(a == b) for a, b in zip(preds, mnist['labels']))}/{len(preds)}"
        if "reference_predictions" in mnist:
            mnist["agree_with_reference_plain_model"] = f"{sum(int(a == b) for a, b in zip(preds, mnist['reference_predictions']))}/{len(preds)}"
        check["mnist"] = mnist
    streamed = None
    if modes:
        streamed = r.get("streamed")
        for sm in streamed or []:
            sm["vs_resident"] = round(sm["images_per_s"] / r["images_per_s"], 4)
            if sm["mode"] == "plaintext":                            # fresh device-side encryptions of the same images: checked by decrypting what came back
                chk_pt, ok_pt = client.verify(cfg_name, imgs, outs_pt, None, golden=False)
                sm["check"] = {k_: chk_pt[k_] for k_ in ("predictions_match_plain_model", "max_logit_abs_err", "noise_budget_bits")}
                ok = ok and ok_pt
            else:
                ok = ok and bool(sm["outputs_identical_to_resident"])
    check["last_timed_launch_identical_to_first"] = r.get("last_timed_launch_identical_to_first")
    ok = ok and bool(r.get("last_timed_launch_identical_to_first"))
    verified = ranks.count(ok) if ranks else int(ok)
    check["ranks_verified"] = f"{verified}/{world}"
    ok = verified == world
    if ranks and rank != 0:
        return None, ok

    # ---- per-layer figures and the roofline of the dominant kernel: algorithmic bytes per launch / measured launch duration
    L = r["layers"]
    reenc_at = r.get("layer_before_reenc", -1)                  # (index into the FUSED network's layers; -1: no refresh)
    ms_per_layer = {}
    for i, (nm, ms) in enumerate(zip(L, r["T_LAYER_ms_per_image"])):
        if i == reenc_at:
            ms_per_layer["T_REENC"] = r["T_REENC_ms_per_image"]     # the client-side refresh, where mainparams.cpp:81 puts its column
        ms_per_layer[nm] = ms
    kernels = {nm: kn for nm, kn in zip(L, r["kernel_per_layer"]) if kn}
    plan = None if args.no_fuse else geometry.fused_plan(client.E, cfg["model"], names=L)
    roofline, useful, layer_gbps = None, {}, {}
    if plan is not None and [pl[1] for pl in plan] == L:
        split = next((i for i, pl in enumerate(plan) if pl[0] == "fc" and i > 0), len(plan)) if G > 1 else len(plan)
        per_launch = lambda i: C * G if i >= split else C          # images per Layer::forward call (two-level chunking: the dense layers once per group)
        for i, (kind, name, a, ish, osh) in enumerate(plan):
            if kind in ("conv", "fc") and kernels.get(name, "").startswith("mfma_mac2w_kernel"):
                useful[name] = round(1.0 / geometry.limb_exec_over_useful(kind, a, per_launch(i), int(np.prod(osh))), 4)
        for i, (kind, name, a, ish, osh) in enumerate(plan):        # every layer call against HBM: algorithmic bytes / its measured duration
            if r["layer_launch_ms"][i] > 0:
                layer_gbps[name] = round(geometry.layer_bytes_and_macs(client.E, kind, a, ish, osh, per_launch(i))[0] / (r["layer_launch_ms"][i] * 1e-3) / 1e9, 1)
        dom = int(np.argmax(r["T_LAYER_ms_per_image"]))
        kind, name, a, ish, osh = plan[dom]
        CL, dur_ms = per_launch(dom), r["layer_launch_ms"][dom]
        alg_bytes, macs = geometry.layer_bytes_and_macs(client.E, kind, a, ish, osh, CL)
        achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        kname = kernels.get(name, "").split(" ")[0] or kind
        label = f"{kname} ({name}, {CL} images/launch)" if kind in ("conv", "fc") else "Square + relinearise sequence (crc_square_relin_forms)" if kind == "square" else "Square + pooled key switch sequence (crc_square_pool_relin_forms)" if kind == "squarepool" else f"{kind} ({name})"
        traffic, traffic_source = geometry.offline_traffic(cfg_name, kind, label, CL * int(np.prod(ish)))
        modmul_s = macs * 2 * cfg["k"] * cfg["n"] / (dur_ms * 1e-3) if dur_ms > 0 and macs else None
        if kname == "mfma_mac2w_kernel" and modmul_s:
            # bound by the int8 MFMA rate, not by HBM: 49 limb products (98 int8 operations) per modular multiply-add against the dense int8 peak (2x the bf16 rate per
            # clock: MI355X_MICROARCH.md, matrix cores).  The HBM view of the same launch stays beside it
            tops = modmul_s * 98 / 1e12
            eou = geometry.limb_exec_over_useful(kind, a, CL, int(np.prod(osh)))
            roofline = dict(bound="mfma", achieved=round(tops * eou, 1), peak=INT8_PEAK_TOPS, unit="TOP/s (int8)", frac=round(tops * eou / INT8_PEAK_TOPS, 5), useful_achieved=round(tops, 1),
                            useful_frac=round(tops / INT8_PEAK_TOPS, 5), traffic=traffic,
                            ops="int8 multiply and add, 98 per modular multiply-add (7 x 7 balanced base-256 limb products); useful = the layer's ct x pt multiply-adds x 2 polys x k n, "
                                "executed = the same with rows / reduction steps / filters padded to the kernel's tiles",
                            hbm_achieved_GBps=round(achieved, 2), hbm_frac=round(achieved / HBM_PEAK_GBS, 5))
        else:
            roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic)
        roofline.update(traffic_source=traffic_source, kernel=label, launch_ms=round(float(dur_ms), 3), launches_timed=r["layer_launches"][dom],
                        kernel_timing="HIP events on the launch stream around the layer call, inside the timed region, averaged over its launches (Network::time_with_events); the call is "
                                      "the kernel named plus, where the neighbouring layer wants another operand form, one conversion kernel",
                        algorithmic_bytes_per_launch=int(alg_bytes), modmul_per_s=round(modmul_s, 1) if modmul_s else None)
    # ---- the reference's published run of this configuration (thesis: one image on a 40-core Xeon), column by column; fused layers take the sum of their columns
    published = None
    if cfg.get("published") and not args.no_fuse:
        from crcnn_amd.netrun import TOPOLOGIES
        pub = cfg["published"]
        cols = [c_ for c_ in pub["columns"] if c_ != "T_REENC"]
        col_of = {nm: (cols[i], pub["seconds"][pub["columns"].index(cols[i])]) for i, (_, nm, _) in enumerate(TOPOLOGIES[cfg["model"]])}
        rows_ = []
        for nm, ms in ms_per_layer.items():
            if nm == "T_REENC":
                rows_.append(dict(columns="T_REENC", layer="refresh (decrypt + re-encrypt, network.cpp:30-34)", published_s=pub["seconds"][pub["columns"].index("T_REENC")], ours_ms=ms))
            else:
                parts = [col_of[p_] for p_ in nm.split("+")]
                rows_.append(dict(columns="+".join(c_ for c_, _ in parts), layer=nm, published_s=round(sum(s_ for _, s_ in parts), 2), ours_ms=ms))
        for rw in rows_:
            rw["ratio"] = round(rw["published_s"] * 1e3 / rw["ours_ms"], 1) if rw["ours_ms"] > 0 else None
        ours_s = 1.0 / r["images_per_s"]
        published = dict(source=pub["source"], published_s_per_image=pub["total_s"], ours_s_per_image=round(ours_s, 6), vs_published=round(pub["total_s"] / ours_s, 1),
                         refresh_share_of_image=round(r["T_REENC_ms_per_image"] / (ours_s * 1e3), 4), per_column=rows_,
                         note="published: single-image latency on the thesis' CPU; ours: per-image time inside a batch on one MI355X (throughput), same parameters, same refresh point, "
                              "decrypted outputs equal to the compiled reference's (check.golden_match)")
    cpu_line = None
    if with_cpu and args.cpu_seconds > 0 and world == 1:           # (the CPU baseline is timed on rank 0 at N = 1 only)
        evk = client.evaluation_keys() if any(pl[0] in ("square", "squarepool") for pl in (plan or [])) else None
        # (--cpu-seconds >= 600 buys ONE WHOLE image through the compiled reference instead of the sampled pieces: minutes for PlainModelTiny at n = 4096 on 16 cores)
        cpu_line = (cpu.cpu_baseline_reference(cfg, q, client.W, x0, cores, evk=evk, full_image=args.cpu_seconds >= 600) or
                    cpu.cpu_baseline(cfg, q, client.W, x0, args.cpu_seconds))
        cpu_line["value"] = round(cpu_line["value"], 6); cpu_line["mac_per_s"] = round(cpu_line["mac_per_s"], 1)
        c1_path = os.path.join(ROOT, "tests", "golden", "c1_tiny4096_t32.json")
        if cfg_name == "tiny4096" and os.path.exists(c1_path):      # configs[0] measured in full in the build container (not extrapolated): oracle/make_c1.py
            c1f = json.load(open(c1_path))
            full_ips = c1f.get("images_per_s_adjusted", c1f["images_per_s"])
            cpu_line["c1_in_full"] = dict(images=len(c1f["images"]), images_per_s=full_ips, total_wall_s=c1f.get("total_wall_s_adjusted", c1f["total_wall_s"]),
                                          threads=c1f["ref_threads"], where="build container (8 cores), the compiled reference on 32 images: tests/golden/c1_tiny4096_t32.json")
            # both figures side by side: the one measured in full (other machine, 8 cores) and this run's sample scaled to the same core count
            cpu_line["value_measured_in_full_elsewhere"] = dict(images_per_s=full_ips, cores=8, per_core=round(full_ips / 8, 7), threads=c1f["ref_threads"])
            cpu_line["value_per_core"] = round(cpu_line["value"] / max(1, cpu_line["cores"]), 7)
    hbm = r["hbm"]
    line = {
        "metric": "encrypted images/sec", "value": round(r["images_per_s"], 4), "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": f"synthetic ({D} distinct MNIST-like encrypted images tiled to the batch: one launch of {C * G} images resident in HBM, re-read by every launch of the step; "
                f"trained weights from {cfg['model']}.h5)",
        # (the batch actually evaluated per step: whole launches of C * G images -- 1020, not 1024, for wopad16384's launches of 30)
        "config": {"workload": f"{cfg['model']}.h5 n={cfg['n']} k={cfg['k']} t=2^{cfg['t'].bit_length() - 1} batch={r.get('batch', B)}/GPU chunk={C}" + (f" (dense layers: {C * G})" if G > 1 else "") + f" ({cfg_name}, BASELINE configs)",
                   "host": r["host"], "mode": "NTT-resident" + ("" if args.no_fuse else " + Network::fuse (conv/pool and batch-norm folding)") +
                   ("" if r.get("matrix_cores", True) else "; matrix_cores = false: every conv / dense layer on the vector-ALU kernel (north_star's no-MFMA path)") +
                   (f"; client-side refresh in front of layer {cfg['reenc']} on the device (crc_refresh_dev)" if "reenc" in cfg else ""), "parallelism": f"image-sharded x{world}"},
        "ms_per_layer": ms_per_layer, "mac_kernel_per_layer": kernels, "mfma_useful_frac_per_layer": useful, "layer_hbm_GBps": layer_gbps, "roofline": roofline, "cpu_baseline": cpu_line,
        "check": dict(check, all_ok=bool(ok)), "vs_published": published,
        "setup_s": r["setup_s"], "client_setup_s": round(client_s, 1), "weight_broadcast": r.get("weight_broadcast"), "timing": r.get("timing"), "per_rank": r.get("per_rank"),
        "streamed": streamed, "ms_per_image_sync_each": r.get("ms_per_image_sync_each"),
        "hbm_plan": dict(hbm, note="bytes on this rank: parameters = the layers' encoded weights in their kernels' operand forms (limb copies replace the canonical ones), activation_slots = "
                                   "the two ping-pong tensors Network::forward keeps + the dense layers' group input, work_buffer = the shared scratch of the layer calls, input_launch = the "
                                   "encrypted images of one launch"),
    }
    return line, bool(ok)


def twin_line(args, D_, cfg_name, steps, warmup, batch, full):
    from benchkit import twin
    return twin.run_config(args, D_, cfg_name, steps, warmup, batch=batch, chunk=args.chunk, full=full)


ALSO_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "data", "ms_per_layer", "mac_kernel_per_layer", "mfma_useful_frac_per_layer", "layer_hbm_GBps",
             "roofline", "cpu_baseline", "check", "vs_published", "setup_s", "weight_broadcast", "per_rank", "streamed", "hbm_plan", "python_twin")

# what the default invocation measures beside the tiny4096 headline, in this order (configurations that share a parameter set follow each other: they share the
# encrypted inputs), with the steps / batch of each: the BASELINE configs ([2], [4]'s per-GPU share), CrCNN's own coefficient moduli (k = 4 at n = 8192, all eight primes at n = 16384: fc3 streams its weights), north_star's
# no-MFMA path, and the reference's two PUBLISHED configurations with their client-side refresh
AUTO_ALSO = [("tiny4096_valu", dict(steps=2)), ("approx8192", dict()), ("wopad16384", dict()), ("approx8192k4", dict(steps=1, batch=512, distinct=8, cpu=False)),
             ("wopad16384k8", dict(steps=1, distinct=4, cpu=False)), ("tiny2048r", dict(steps=2, cpu=False)), ("approx4096r", dict(steps=2, cpu=False))]
# single-image latency (mainparams.cpp:85-112: the reference's whole usage model is one image at a time): bench_host batch=1 chunk=1, a synchronisation per image
LATENCY_STEPS = {"tiny4096": 20, "approx8192": 20, "wopad16384": 5}


def latency_line(args, cfg_name, device):
    """one encrypted image at a time through the same network: Network::forward on a batch of ONE, the stream synchronised after every image.  Reports the latency,
    the per-layer times, the multiply-accumulate kernel crc_plan_mac picks per layer at two rows per launch, and every layer call against the HBM roofline -- at one
    image a dense layer is a pure weight stream (PlainModelTiny's fc3: 34 GB of NTT-form weights for 2 x 512 outputs)"""
    ln, ok = run_host(args, cfg_name, LATENCY_STEPS.get(cfg_name, 10), 2, batch=1, device=device, over=dict(chunk=1, tail=1, distinct=1), extra=dict(sync_each=1), with_cpu=False)
    lat = dict(workload=ln["config"]["workload"], latency_ms=ln["ms_per_image_sync_each"], images_timed=ln["steps"], ms_per_layer=ln["ms_per_layer"],
               kernel_per_layer=ln["mac_kernel_per_layer"], layer_hbm_GBps=ln["layer_hbm_GBps"],
               layer_hbm_frac={k_: round(v / HBM_PEAK_GBS, 4) for k_, v in ln["layer_hbm_GBps"].items()}, check=ln["check"], setup_s=ln["setup_s"],
               how="bench_host batch=1 chunk=1 sync_each=1: wall clock around Network::forward + stream synchronisation per image, inputs resident; per-layer times from HIP events")
    return lat, ok


def main():
    args = parse()
    from benchkit import dist as bdist
    if args.launch_check:
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            sys.exit(bdist.self_launch(args))
        sys.exit(bdist.launch_check(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain `python bench.py --gpus N`: start the ranks ourselves
        sys.exit(bdist.self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    also = args.also
    if also == "auto":
        # the default invocation also measures BASELINE configs[2] (ApproxPlainModel, n = 8192, k = 3) and configs[4]'s per-GPU share (PlainModelWoPad, n = 16384, k = 4,
        # 1024 images) in the same process
        # (N > 1: the two BASELINE configurations that are quoted on eight GPUs -- configs[3] and configs[4] -- and nothing else: the single-GPU lines would only make
        # every rank encrypt more inputs on its share of the host cores)
        auto_names = [nm for nm, _ in AUTO_ALSO] if world == 1 else ["approx8192", "wopad16384"]
        also = ",".join(auto_names) if args.config == "tiny4096" and args.batch is None else "none"
    names = [] if also == "none" else also.split(",")
    auto = dict(AUTO_ALSO) if args.also == "auto" else {}
    latency_on = args.latency == "on" or (args.latency == "auto" and args.also == "auto" and args.config == "tiny4096" and args.batch is None and world == 1)
    if args.stream_inputs == "auto":
        # (N = 1 only: eight ranks streaming at once would pin ~10 GB of host memory each and share one host's PCIe root -- a figure about the host, not the engine)
        args.stream_inputs = "both" if args.config == "tiny4096" and args.batch is None and world == 1 else "none"
    line, ok = None, True
    if world == 1 or not args.python_twin:
        ranks = bdist.HostRanks(args) if world > 1 else None
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if ranks and os.environ.get("CRC_COMM_TRANSPORT") == "shm":  # a rehearsal with more ranks than GPUs shares devices; the KFD topology is read, no HIP call is made
            local %= max(1, bdist.gpu_count_without_hip())
        line, ok = run_host(args, args.config, args.steps, args.warmup, batch=args.batch, device=local, ranks=ranks, stream=args.stream_inputs)
        if args.python_twin and world == 1:
            D_ = bdist.Dist(args)
            tw, ok2 = twin_line(args, D_, args.config, args.steps, args.warmup, args.batch, True)
            line["python_twin"] = dict(value=tw["value"], ms_per_layer=tw["ms_per_layer"], check=tw["check"], vs_host=round(tw["value"] / line["value"], 4),
                                       reference_layer_structure=tw.get("reference_layer_structure"))
            ok = ok and ok2
        latency = []
        if latency_on and (args.config in LATENCY_STEPS or args.latency == "on"):
            lat, ok2 = latency_line(args, args.config, local); latency.append(lat); ok = ok and ok2
        for nm in names:
            o = auto.get(nm, {})
            second, ok2 = run_host(args, nm, o.get("steps", args.also_steps), 0, batch=args.also_batch or o.get("batch"), device=local, ranks=ranks,
                                   stream="ciphertext" if args.stream_inputs != "none" and nm == "approx8192" else "none",
                                   over=dict(distinct=o.get("distinct")), with_cpu=o.get("cpu", True))
            ok = ok and ok2
            if line is not None:
                line.setdefault("also", []).append({k_: second[k_] for k_ in ALSO_KEYS if k_ in second})
            if latency_on and nm in LATENCY_STEPS:
                lat, ok2 = latency_line(args, nm, local); latency.append(lat); ok = ok and ok2
        drop_inputs()
        if line is not None:
            if latency:
                line["latency"] = latency
            # the driver's record keeps the contract keys, `config`, `roofline` and `cpu_baseline` whole: the figures a reader should not have to dig out of "also" /
            # "streamed" / "latency" are summarised inside `config`
            cfgd = line["config"]
            if line.get("streamed"):
                cfgd["streamed_images_per_s"] = {sm["mode"]: dict(images_per_s=round(sm["images_per_s"], 2), vs_resident=sm["vs_resident"], h2d_GBps=sm["h2d_GBps"]) for sm in line["streamed"]}
            if latency:
                cfgd["latency_ms_single_image"] = {lt["workload"].split(" (")[-1].split(",")[0]: lt["latency_ms"] for lt in latency}
            if line.get("also"):
                cfgd["also_summary"] = [dict(workload=a_["config"]["workload"], images_per_s=a_["value"], golden_match=a_["check"].get("golden_match"), all_ok=a_["check"].get("all_ok"),
                                             roofline_frac=(a_.get("roofline") or {}).get("frac"), vs_published=(a_.get("vs_published") or {}).get("vs_published"),
                                             streamed=[dict(mode=sm["mode"], images_per_s=round(sm["images_per_s"], 2), vs_resident=sm["vs_resident"]) for sm in (a_.get("streamed") or [])] or None)
                                        for a_ in line["also"]]
        if ranks:
            ranks.close()
    else:
        D_ = bdist.Dist(args)
        line, ok = twin_line(args, D_, args.config, args.steps, args.warmup, args.batch, True)
        for nm in names:
            second, ok2 = twin_line(args, D_, nm, args.also_steps, 0, args.also_batch, False)
            ok = ok and ok2
            if line is not None:
                line.setdefault("also", []).append({k_: second[k_] for k_ in ALSO_KEYS if k_ in second})
        D_.dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)
    if not ok:
        sys.stderr.write("bench.py: a self-check FAILED (see \"check\"): the reported value is not a valid measurement\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
