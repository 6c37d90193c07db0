#!/usr/bin/env python3
"""Copy the outputs of tools/measure_round.sh (gpurun_out/final) into profiles/ under this round's names and derive the two small
summaries the docs cite: PMC traffic of the dominant launch and the MAC launches grouped by grid size.  usage: collect_profiles.py r01"""
import collections, csv, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final"); P = os.path.join(ROOT, "profiles")
def cp(src, dst): shutil.copy(os.path.join(F, src), os.path.join(P, f"{R}_{dst}"))
for c in ("tiny4096", "approx8192", "wopad16384_b96"):
    cp(f"bench_{c}.json", f"bench_{c}.json")
cp("prof_tiny/tiny_kernel_stats.csv", "bench_tiny4096_kernel_stats.csv")
cp("prof_approx/approx_kernel_stats.csv", "bench_approx8192_b96_kernel_stats.csv")
for src, dst in (("ntt_elementwise.txt", "ntt_elementwise_kernels.txt"), ("square.txt", "square_relin.txt"), ("mac_geometries.txt", "mac_geometries.txt")):
    lines = [l for l in open(os.path.join(F, src)) if "amdgpu.ids" not in l]
    open(os.path.join(P, f"{R}_{dst}"), "w").writelines(lines)
# PMC: counters are KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md)
def tot(path, counter):
    acc = collections.defaultdict(float); t = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "mac" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"]); t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    k = sorted(acc)[-1]
    return acc[k], t[k]
f, ms = tot(os.path.join(F, "pmc_fetch/f_counter_collection.csv"), "FETCH_SIZE"); w, _ = tot(os.path.join(F, "pmc_write/w_counter_collection.csv"), "WRITE_SIZE")
d = {"tiny4096": {"kernel": "mac3_kernel (pool2_features.conv2+pool2, 24 images/launch)", "fetch_bytes": f * 1024 * 2, "write_bytes": w * 1024, "traffic_bytes": f * 2048 + w * 1024,
                  "launch_ms_under_pmc": round(ms, 3),
                  "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/measure_round.sh) on tools/bench_mac.py conv2p 24 = the fused conv2+pool2 launch of "
                          "the bench at chunk 24; counters are KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of coalesced 8/16-B-per-lane reads: MI355X_MICROARCH.md; calibrated "
                          "this round on ntt_rows_kernel reading exactly 1 GiB -> 524 611 KiB)"}}
json.dump(d, open(os.path.join(P, f"{R}_pmc_traffic.json"), "w"), indent=1)
g = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(F, "prof_tiny/tiny_kernel_trace.csv"))):
    if "mac" in r["Kernel_Name"] and "_kernel<" in r["Kernel_Name"]:
        g[(int(r["Grid_Size_X"]), r["Kernel_Name"].split("(")[0][5:])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(P, f"{R}_bench_tiny4096_mac_by_launch_shape.txt"), "w") as o:
    o.write("# rocprofv3 --kernel-trace of `python3 bench.py --cpu-seconds 0 --unfused-images 48` (same run as the kernel_stats.csv next to this file):\n"
            "# conv/dense launches grouped by grid size (threads); the fused conv2+pool2 launch at 24 images (bench.py roofline.kernel) is grid 16777216\n")
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        o.write(f"grid={k[0]:>10d} {k[1]:34s} launches={len(v):4d} avg={sum(v) / len(v):9.3f} ms min={min(v):9.3f} max={max(v):9.3f} total={sum(v):10.1f} ms\n")
print(json.dumps(d["tiny4096"]))
