#!/usr/bin/env python3
"""Copy the outputs of tools/measure_round6.sh (gpurun_out/final6: one box, one call) into profiles/r06_* and derive (a) the agreement between bench.py's HIP-event
launch time of the headline kernel and rocprofv3's kernel-stat average of the same kernel on the same box, (b) the PMC traffic summary bench.py cites
(roofline.traffic), on the kernel names of the final tree, (c) the batch-1 weight stream's FETCH_SIZE against its algorithmic bytes."""
import collections, csv, glob, json, os, shutil, sys
R = "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final6"); P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(os.path.join(F, src)):
        shutil.copy(os.path.join(F, src), os.path.join(P, f"{R}_{dst}")); return True
    return False


cp("bench_driver_invocation.json", "bench_driver_invocation_final.json"); cp("bench_driver_invocation.time", "bench_driver_invocation_final.time")
for cfg in ("tiny4096", "approx8192", "wopad16384", "tiny4096_b1", "approx8192_b1"):
    for f in glob.glob(os.path.join(F, f"prof_{cfg}", "**", f"{cfg}_kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, f"{R}_bench_{cfg}_kernel_stats.csv"))
for tag in ("8192_3_1250", "16384_4_1250"):
    cp(f"prof_square_pool_{tag}.txt", f"square_pool_{tag}_kernels_final.txt")
for src, dst in (("ntt_elementwise.txt", "ntt_elementwise_kernels.txt"), ("square_pool.txt", "square_pool_one_key_switch.txt"), ("device_encryptor.txt", "device_encryptor.txt")):
    if os.path.exists(os.path.join(F, src)):
        open(os.path.join(P, f"{R}_{dst}"), "w").writelines(l for l in open(os.path.join(F, src)) if "amdgpu.ids" not in l and not l.startswith("+"))


def stats(cfg):
    f = os.path.join(P, f"{R}_bench_{cfg}_kernel_stats.csv")
    return {r["Name"]: r for r in csv.DictReader(open(f))} if os.path.exists(f) else {}


# (a) the headline kernel: bench line (HIP events inside the timed region) against the profiler's average, same box
try:
    line = json.loads(open(os.path.join(F, "bench_driver_invocation.json")).read().strip().splitlines()[-1])
    rows = []
    for cfg, ln in [("tiny4096", line)] + [(a["config"]["workload"].split(" (")[-1].split(",")[0], a) for a in line.get("also", [])]:
        st = stats(cfg)
        rf = ln.get("roofline") or {}
        if not st or not rf:
            continue
        if rf.get("kernel", "").startswith("mfma_mac2w_kernel"):
            cand = {k: v for k, v in st.items() if "mfma_mac2w_kernel" in k}
            name = max(cand, key=lambda k: float(cand[k]["AverageNs"]))        # the dominant layer's instance is the longest one
            rows.append(dict(config=cfg, kernel=name, bench_launch_ms=rf["launch_ms"], rocprof_average_ms=round(float(cand[name]["AverageNs"]) / 1e6, 3), rocprof_calls=int(cand[name]["Calls"]),
                             note="the layer call bench.py times is this kernel plus, where the neighbouring layer wants another operand form, one conversion kernel"))
        else:                                                                     # a sequence of kernels (Square + pooled key switch): sum of their per-call averages x calls per layer call
            seq = [k for k in st if any(t in k for t in ("sq64_", "relin_", "ntt_rows_wave_kernel<true, 4", "ntt_rows_wave_kernel<true, 5", "ntt_rows_kernel<true, true, 4", "ntt_rows_kernel<true, true, 5"))]
            tot = sum(float(st[k]["TotalDurationNs"]) for k in seq) / 1e6
            rows.append(dict(config=cfg, kernel=rf.get("kernel"), bench_launch_ms=rf["launch_ms"], bench_launches=rf.get("launches_timed"), rocprof_total_ms_of_the_sequence_kernels=round(tot, 3),
                             kernels=seq))
    for r_ in rows:
        if "rocprof_average_ms" in r_:
            r_["rocprof_over_bench"] = round(r_["rocprof_average_ms"] / r_["bench_launch_ms"], 4)
    json.dump(dict(box="one gpurun call: tools/measure_round6.sh (bench, then rocprofv3 --kernel-trace --stats over the same bench_host command lines with steps=3)", rows=rows),
              open(os.path.join(P, f"{R}_headline_kernel_agreement.json"), "w"), indent=1)
    print(json.dumps(rows, indent=1)[:1500])
except Exception as e:
    print("no agreement summary:", e)


# PMC: counters are KiB; FETCH_SIZE is doubled on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section)
def tot(path, counter, kernel, pick="last"):
    acc = collections.defaultdict(float); t = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"]); t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    k = sorted(acc, key=int)[-1] if pick == "last" else max(acc, key=lambda d: acc[d])
    return acc[k], t[k]


def cc(sub):
    return glob.glob(os.path.join(F, sub, "**", "*counter_collection.csv"), recursive=True)[0]


out = {"_round": "every entry measured on the final tree of round 6 by tools/measure_round6.sh (one box, one call); kernel names are the current ones"}
try:
    d = {}
    for kern in ("mfma_mac2w_kernel", "limb_pack_tensor_kernel", "slotmajor_to_limb_kernel"):
        try:
            f, ms = tot(cc("pmc_fetch"), "FETCH_SIZE", kern); w, _ = tot(cc("pmc_write"), "WRITE_SIZE", kern)
        except (IndexError, ValueError):
            continue
        d[kern] = dict(fetch_bytes=f * 2048, write_bytes=w * 1024, traffic_bytes=f * 2048 + w * 1024, launch_ms_under_pmc=round(ms, 3))
    out["tiny4096"] = {"kernel": "mfma_mac2w_kernel (pool2_features.conv2+pool2, 128 images/launch)", **d["mfma_mac2w_kernel"],
                       "other_kernels_of_the_layer_call": {k: v for k, v in d.items() if k != "mfma_mac2w_kernel"},
                       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes on `tools/bench_mac.py conv2p 128 1 limbk` = the conv2+pool2 launch of the bench at chunk 128 on a "
                               "limb-form input; counters are KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of coalesced 16-B-per-lane reads, global_load and LDS-DMA alike: "
                               "MI355X_MICROARCH.md).  Algorithmic bytes of the launch in limb form: 67.6 GB tensor + 4.2 GB weights + 15.0 GB result (fc3's limb tensor, written by the kernel itself) = 86.8 GB"}
    try:
        sq = {}
        for cn in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"):
            sq[cn], ms = tot(cc("pmc_sq"), cn, "mfma_mac2w_kernel")
        g, msg = tot(cc("pmc_grbm"), "GRBM_GUI_ACTIVE", "mfma_mac2w_kernel")
        out["tiny4096"]["issue_split"] = {**sq, "launch_ms_under_pmc": round(ms, 3), "GRBM_GUI_ACTIVE": g, "effective_clock_GHz": round(g / 8 / (msg * 1e-3) / 1e9, 3)}
    except Exception as e:
        print("no issue split:", e)
    try:
        c1 = {}
        for kern in ("mfma_conv1_kernel", "limb_pack_rows1_kernel"):
            f, ms = tot(cc("pmc_c1_fetch"), "FETCH_SIZE", kern); w, _ = tot(cc("pmc_c1_write"), "WRITE_SIZE", kern)
            c1[kern] = dict(fetch_bytes=f * 2048, write_bytes=w * 1024, traffic_bytes=f * 2048 + w * 1024, launch_ms_under_pmc=round(ms, 3))
        out["tiny4096_conv1"] = {"kernel": "mfma_conv1_kernel (pool1_features.conv1+pool1, 128 images/launch, limb-tensor output)", **c1["mfma_conv1_kernel"], "limb_pack_rows1_kernel": c1["limb_pack_rows1_kernel"],
                                 "note": "algorithmic: 13.2 GB limb images in, 67.6 GB limb tensor out (mfma_conv1_kernel); 13.2 GB NTT-form images in, 13.2 GB limb images out (limb_pack_rows1_kernel)"}
    except Exception as e:
        print("no conv1 summary:", e)
except Exception as e:
    print("no headline PMC summary:", e)
for tag, cn in (("8192_3_1250", "approx8192"), ("16384_4_512", "wopad16384")):
    try:
        sm = json.load(open(os.path.join(F, f"pmc_square_{tag}.json")))
        out[cn] = dict(kernel=f"Square + relinearise sequence (crc_square_relin_forms, n = {sm['config']['n']}, k = {sm['config']['k']})", per_ciphertext=True,
                       traffic_bytes_per_ciphertext=sm["total"]["sum"], read_bytes_per_ciphertext=sm["total"]["read"], write_bytes_per_ciphertext=sm["total"]["write"],
                       algorithmic_bytes_per_ciphertext=sm["total"]["algorithmic"], ratio_to_algorithmic=sm["total"]["ratio"], kernels=sm["kernels"], read_counter_calibration=sm["read_counter_calibration"])
    except Exception as e:
        print("no square PMC summary for", tag, e)
for tag, cn in (("8192_3_1250", "approx8192"), ("16384_4_1250", "wopad16384")):
    try:
        sm = json.load(open(os.path.join(F, f"pmc_square_pool_{tag}.json")))
        out.setdefault(cn, {})["pooled"] = dict(
            kernel=f"Square + pooled key switch sequence (crc_square_pool_relin_forms, n = {sm['config']['n']}, k = {sm['config']['k']}, 5 x 5 -> 4 x 4)",
            traffic_bytes_per_ciphertext=sm["total"]["sum"], read_bytes_per_ciphertext=sm["total"]["read"], write_bytes_per_ciphertext=sm["total"]["write"],
            algorithmic_bytes_per_ciphertext=sm["total"]["algorithmic"], ratio_to_algorithmic=sm["total"]["ratio"], kernels=sm["kernels"], read_counter_calibration=sm["read_counter_calibration"],
            note="per SQUARED ciphertext (1250 per image in, 800 out)")
        shutil.copy(os.path.join(F, f"pmc_square_pool_{tag}.json"), os.path.join(P, f"{R}_pmc_square_pool_{tag}_final.json"))
    except Exception as e:
        print("no pooled square PMC summary for", tag, e)
# (c) the batch-1 weight stream: fc3 of PlainModelTiny on ONE image = 1024 x 512 weights of 2 x 4096 residues, each read once
try:
    f, ms = tot(cc("pmc_b1_fetch"), "FETCH_SIZE", "mac_stream_kernel", pick="max"); w, _ = tot(cc("pmc_b1_write"), "WRITE_SIZE", "mac_stream_kernel", pick="max")
    alg_w = 1024 * 512 * 2 * 4096 * 8; alg_x = 1024 * 2 * 2 * 4096 * 8; alg_y = 512 * 2 * 2 * 4096 * 8
    b1 = dict(kernel="mac_stream_kernel<4, 2, ..> (classifier.fc3 of PlainModelTiny, n = 4096, k = 2, ONE image: bench_host batch=1 chunk=1)", fetch_bytes=f * 2048, write_bytes=w * 1024,
              launch_ms_under_pmc=round(ms, 3), algorithmic_bytes=dict(weights=alg_w, input_ciphertexts=alg_x, output_ciphertexts=alg_y, total=alg_w + alg_x + alg_y),
              fetch_over_algorithmic_reads=round(f * 2048 / (alg_w + alg_x), 3), achieved_GBps_under_pmc=round((alg_w + alg_x + alg_y) / (ms * 1e-3) / 1e9, 1),
              frac_of_8TBps=round((alg_w + alg_x + alg_y) / (ms * 1e-3) / 8e12, 4),
              note="FETCH_SIZE in KiB, doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane coalesced reads on gfx950; the x rows are re-read by every filter group "
                   "(128 groups of 4 filters) but stay in L2 / MALL: fetch_over_algorithmic_reads says how much of that reaches HBM")
    json.dump(b1, open(os.path.join(P, f"{R}_fc3_b1_pmc.json"), "w"), indent=1)
    out["tiny4096_fc3_b1"] = b1
    print(json.dumps(b1)[:600])
except Exception as e:
    print("no batch-1 PMC summary:", e)
json.dump(out, open(os.path.join(P, f"{R}_pmc_traffic.json"), "w"), indent=1)
print("wrote", os.path.join(P, f"{R}_pmc_traffic.json"), "entries", [k for k in out if not k.startswith("_")])
