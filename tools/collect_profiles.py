#!/usr/bin/env python3
"""Copy the outputs of tools/measure_round.sh (gpurun_out/final) into profiles/ under this round's names and derive the PMC traffic summary of the
dominant launch that bench.py cites (roofline.traffic).  usage: collect_profiles.py r02"""
import collections, csv, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final"); P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(os.path.join(F, src)):
        shutil.copy(os.path.join(F, src), os.path.join(P, f"{R}_{dst}"))


for c in ("default_invocation", "approx8192k4_b256", "wopad16384k8_b96", "tiny4096_with_python_twin"):
    cp(f"bench_{c}.json", f"bench_{c}.json")
cp("prof_tiny4096/tiny4096_kernel_stats.csv", "bench_tiny4096_b256_kernel_stats.csv")
cp("prof_approx8192/approx8192_kernel_stats.csv", "bench_approx8192_b96_kernel_stats.csv")
cp("prof_c1/c1_kernel_stats.csv", "conv1_4096k2_b128_kernel_stats.csv")
for tag in ("8192_3_1250", "16384_4_512", "16384_8_256"):
    cp(f"prof_square_{tag}.txt", f"square_relin_{tag}_kernels.txt"); cp(f"prof_square_old_{tag}.txt", f"square_relin_{tag}_kernels_round2_path.txt")
cp("square_paths.txt", "square_relin_paths.txt")
for tag in ("8192_3_1250", "16384_4_1250"):           # the Square + pooling pair with one key switch per pooled ciphertext (per squared ciphertext)
    cp(f"prof_square_pool_{tag}.txt", f"square_pool_{tag}_kernels.txt")
cp("square_pool.txt", "square_pool_one_key_switch.txt")
# one file with the before / after per-kernel numbers and the path-by-path timings (what VERDICT r2 item 1 asks for)
with open(os.path.join(P, f"{R}_square_relin.txt"), "w") as f:
    f.write("# Square + relinearise (crc_square_relin_forms, NTT form in and out), us per ciphertext and kernel: rocprofv3 --kernel-trace --stats over tools/bench_square.py\n"
            "# (tools/prof_square.sh; 4 sequences per run).  'round-2 kernels' = CRC_SQ_PATH=1 CRC_RELIN_PATH=1 on the same box and build: SEAL's 61-bit auxiliary base and key\n"
            "# switching over the coefficient moduli; 'this round' = the default (fp64 auxiliary base, key switching over two fp64 primes; round 4: lift fused into the forward\n"
            "# transforms, rows held in registers, gap-1 stage in the fill / drain loops).\n")
    for tag in ("8192_3_1250", "16384_4_512", "16384_8_256"):
        n_, k_, c_ = tag.split("_")
        for src, label in ((f"prof_square_old_{tag}.txt", "round-2 kernels"), (f"prof_square_{tag}.txt", "this round")):
            if os.path.exists(os.path.join(F, src)):
                f.write(f"\n== n = {n_}, k = {k_}, {c_} ciphertexts per call: {label}\n")
                f.writelines(l for l in open(os.path.join(F, src)) if "amdgpu.ids" not in l and not l.startswith("+") and "at::native" not in l and "rocclr" not in l)
    if os.path.exists(os.path.join(F, "square_paths.txt")):
        f.write("\n== wall time per call (HIP events, no profiler), path by path\n")
        f.writelines(open(os.path.join(F, "square_paths.txt")))
cp("../prof_sq_8192_3_1250/sq_8192_3_1250_kernel_stats.csv", "square_relin_8192k3_kernel_stats.csv")
for src, dst in (("ntt_elementwise.txt", "ntt_elementwise_kernels.txt"), ("mac_geometries.txt", "mac_geometries.txt"), ("conv1.txt", "conv1_kernel.txt"), ("mfma_shape.txt", "mfma_shape.txt"),
                 ("square.txt", "square_relin_raw.txt"), ("pack.txt", "limb_pack_weights.txt")):
    if os.path.exists(os.path.join(F, src)):
        lines = [l for l in open(os.path.join(F, src)) if "amdgpu.ids" not in l and not l.startswith("+")]
        open(os.path.join(P, f"{R}_{dst}"), "w").writelines(lines)


# per launch shape: the kernel-stats average mixes launches of different layers (and the single-image module-load pass); the dominant launch is one shape
def by_shape(trace, dst, title):
    if not os.path.exists(trace):
        return
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"].split("(")[0]
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        acc[(name, grid, int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open(dst, "w") as f:
        f.write(f"# {title}, by launch shape (under the profiler the clock is a few % lower than in bench runs)\n# kernel | grid (threads) | workgroup | launches | average ms | min | max\n")
        for (name, grid, wg), v in rows:
            if sum(v) < 0.05:
                continue
            f.write(f"{name[:60]:60s} {grid:>12d} {wg:>5d} {len(v):>5d} {sum(v) / len(v):>10.3f} {min(v):>10.3f} {max(v):>10.3f}\n")


by_shape(os.path.join(F, "prof_tiny4096/tiny4096_kernel_trace.csv"), os.path.join(P, f"{R}_bench_tiny4096_by_launch_shape.txt"),
         "rocprofv3 --kernel-trace of bench.py's measured path (crcnn_amd/lib/bench_host, tiny4096, batch 256, 128 images per launch)")
by_shape(os.path.join(F, "prof_approx8192/approx8192_kernel_trace.csv"), os.path.join(P, f"{R}_bench_approx8192_by_launch_shape.txt"),
         "rocprofv3 --kernel-trace of bench.py's measured path (crcnn_amd/lib/bench_host, approx8192, batch 96, 32 images per launch, dense layers per 64)")

# PMC: counters are KiB; FETCH_SIZE is doubled on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section)
def tot(path, counter, kernel):
    acc = collections.defaultdict(float); t = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"]); t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    k = sorted(acc, key=int)[-1]
    return acc[k], t[k]


out = {}
try:
    d = {}
    for kern in ("mfma_mac2w_kernel", "limb_pack_tensor_kernel", "slotmajor_to_limb_kernel"):
        try:
            f, ms = tot(os.path.join(F, "pmc_fetch/f_counter_collection.csv"), "FETCH_SIZE", kern)
            w, _ = tot(os.path.join(F, "pmc_write/w_counter_collection.csv"), "WRITE_SIZE", kern)
        except IndexError:            # not launched (the GEMM writes a dense consumer's limb tensor itself: no conversion kernel)
            continue
        d[kern] = dict(fetch_bytes=f * 2048, write_bytes=w * 1024, traffic_bytes=f * 2048 + w * 1024, launch_ms_under_pmc=round(ms, 3))
    m = d["mfma_mac2w_kernel"]
    out["tiny4096"] = {"kernel": "mfma_mac2w_kernel (pool2_features.conv2+pool2, 128 images/launch)", **m, "other_kernels_of_the_layer_call": {k: v for k, v in d.items() if k != "mfma_mac2w_kernel"},
                       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/measure_round.sh prof) on `tools/bench_mac.py conv2p 128 1 limbk` = the conv2+pool2 launch "
                               "of the bench at chunk 128 on a limb-form input; counters are KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of coalesced 16-B-per-lane reads, "
                               "global_load and LDS-DMA alike: MI355X_MICROARCH.md).  Algorithmic bytes of the launch in limb form: 67.6 GB tensor + 4.2 GB weights + 15.0 GB result (fc3's limb tensor, written by the kernel itself) = 86.8 GB"}
    # the same for the one-channel convolution (tools/check_conv1.py: last launches = the limb-tensor output of 128 images) and the issue / wait split of mfma_mac_kernel
    try:
        c1 = {}
        for kern in ("mfma_conv1_kernel", "limb_pack_rows1_kernel"):
            f, ms = tot(os.path.join(F, "pmc_c1_fetch/f_counter_collection.csv"), "FETCH_SIZE", kern)
            w, _ = tot(os.path.join(F, "pmc_c1_write/w_counter_collection.csv"), "WRITE_SIZE", kern)
            c1[kern] = dict(fetch_bytes=f * 2048, write_bytes=w * 1024, traffic_bytes=f * 2048 + w * 1024, launch_ms_under_pmc=round(ms, 3))
        out["tiny4096_conv1"] = {"kernel": "mfma_conv1_kernel (pool1_features.conv1+pool1, 128 images/launch, limb-tensor output)", **c1["mfma_conv1_kernel"], "limb_pack_rows1_kernel": c1["limb_pack_rows1_kernel"],
                                 "note": "algorithmic: 13.2 GB limb images in, 67.6 GB limb tensor out (mfma_conv1_kernel); 13.2 GB NTT-form images in, 13.2 GB limb images out (limb_pack_rows1_kernel)"}
        sq = {}
        for cn in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"):
            sq[cn], ms = tot(os.path.join(F, "pmc_sq/s_counter_collection.csv"), cn, "mfma_mac2w_kernel")
        g, msg = tot(os.path.join(F, "pmc_grbm/g_counter_collection.csv"), "GRBM_GUI_ACTIVE", "mfma_mac2w_kernel")
        out["tiny4096"]["issue_split"] = {**{k_: v_ for k_, v_ in sq.items()}, "launch_ms_under_pmc": round(ms, 3), "GRBM_GUI_ACTIVE": g, "effective_clock_GHz": round(g / 8 / (msg * 1e-3) / 1e9, 3),
                                          "note": "wave-parked (s_waitcnt / barrier) = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue stall (MFMA pipe / dependency) = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"}
    except Exception as e:
        print("no conv1 / issue-split summary:", e)
    # Square + relinearise sequence (tools/pmc_square.sh): bytes per ciphertext, keyed by the bench configuration that runs on that ring
    for tag, cfgs in (("8192_3_1250", ("approx8192",)), ("16384_4_512", ("wopad16384",)), ("16384_8_256", ("wopad16384k8",))):
        try:
            sm = json.load(open(os.path.join(F, f"pmc_square_{tag}.json")))
            for cn in cfgs:
                out[cn] = dict(kernel=f"Square + relinearise sequence (crc_square_relin_forms, n = {sm['config']['n']}, k = {sm['config']['k']})", per_ciphertext=True,
                               traffic_bytes_per_ciphertext=sm["total"]["sum"], read_bytes_per_ciphertext=sm["total"]["read"], write_bytes_per_ciphertext=sm["total"]["write"],
                               algorithmic_bytes_per_ciphertext=sm["total"]["algorithmic"], ratio_to_algorithmic=sm["total"]["ratio"], kernels=sm["kernels"],
                               read_counter_calibration=sm["read_counter_calibration"],
                               note="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/bench_square.py (NTT form in and out); the read counter is calibrated on the "
                                    "plain row inverse transform of the same run, whose bytes are known (MI355X_MICROARCH.md gives the factor 2 for 16 B per lane reads; the kernels of the sequence mix widths)")
        except Exception as e:
            print("no square PMC summary for", tag, e)
    # the layer pair Square + pooling with one key switch per pooled ciphertext (CRC_BENCH_SQ_POOL=1): what the fused networks run; bytes per SQUARED ciphertext
    for tag, cfgs in (("8192_3_1250", ("approx8192",)), ("16384_4_1250", ("wopad16384",))):
        try:
            sm = json.load(open(os.path.join(F, f"pmc_square_pool_{tag}.json")))
            for cn in cfgs:
                out.setdefault(cn, {})["pooled"] = dict(
                    kernel=f"Square + pooled key switch sequence (crc_square_pool_relin_forms, n = {sm['config']['n']}, k = {sm['config']['k']}, 5 x 5 -> 4 x 4)",
                    traffic_bytes_per_ciphertext=sm["total"]["sum"], read_bytes_per_ciphertext=sm["total"]["read"], write_bytes_per_ciphertext=sm["total"]["write"],
                    algorithmic_bytes_per_ciphertext=sm["total"]["algorithmic"], ratio_to_algorithmic=sm["total"]["ratio"], kernels=sm["kernels"],
                    note="per SQUARED ciphertext (1250 per image in, 800 out); same passes and calibration as the unpooled sequence")
        except Exception as e:
            print("no pooled square PMC summary for", tag, e)
    json.dump(out, open(os.path.join(P, f"{R}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out["tiny4096"])[:400])
except Exception as e:
    print("no PMC summary:", e)
