#!/usr/bin/env python3
"""Summarise the two PMC passes of tools/pmc_square.sh: HBM bytes per ciphertext and kernel of the Square + relinearise sequence.
Counters are KiB (rocprofv3 derived FETCH_SIZE / WRITE_SIZE).  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B / lane) coalesced
reads and other widths are uncalibrated -- the kernels of this sequence mix widths, so the read counter is calibrated here on a kernel of the same sequence whose bytes are known
exactly: the row inverse transform (ntt_rows_kernel<true, true, 0 / 5>, since round 5 ntt_rows_wave_kernel<true, 5, ..>) reads each of its rows once (8 n bytes per row)."""
import collections, csv, glob, json, sys
O, cfg = sys.argv[1], sys.argv[2]
n, k, cts = [int(v) for v in cfg.split()]
runs = 4                                    # bench_square.py: 1 warm-up + 3 timed sequences


def load(sub, counter):
    f = glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float); calls = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        acc[name] += float(r["Counter_Value"]) * 1024
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[name] += 1
    return acc, calls


fetch, calls = load("fetch", "FETCH_SIZE")
write, _ = load("write", "WRITE_SIZE")
ours = [nm for nm in fetch if not nm.startswith(("void at::", "at::", "__amd"))]
per = cts * runs
cal_name = next((nm for nm in ours if any(t in nm for t in ("ntt_rows_kernel<true, true, 0>", "ntt_rows_kernel<true, true, 5>", "ntt_rows_split_kernel<true, true, 5>",
                                                            "ntt_rows_wave_kernel<true, 5,"))), None)      # (round 4: prologue 5 = the same transform with a scaled result)
cal = None
if cal_name:
    known = 2 * k * 8 * n * per          # the first inverse transform of the NTT-resident input: 2k rows per ciphertext, each read once
    rows_calls = calls[cal_name]
    cal = known / fetch[cal_name] if fetch[cal_name] else None
out = dict(config=dict(n=n, k=k, cts=cts), unit="bytes per ciphertext", read_counter_calibration=dict(kernel=cal_name, factor=cal,
           note="known bytes / FETCH_SIZE of the plain row inverse transform (16 B per lane reads since round 3's pair accesses; the guide's factor for such reads is 2, this kernel's known bytes give the factor used here)"), kernels={})
tot_r = tot_w = 0.0
for nm in sorted(ours, key=lambda x: -(fetch[x] + write.get(x, 0))):
    r = fetch[nm] * (cal or 2.0) / per; w = write.get(nm, 0.0) / per
    tot_r += r; tot_w += w
    out["kernels"][nm[:80]] = dict(read=round(r), write=round(w), launches=calls[nm])
alg = 8 * n * (2 * 2 * k + 2 * 2 * k) // 2      # every distinct operand moved once = 2k rows in + 2k rows out
import os
if os.environ.get("CRC_BENCH_SQ_POOL") == "1":   # Square + pooling pair: 2k rows in per squared ciphertext, 2k rows out per POOLED one (16 of 25)
    alg = 8 * n * 2 * k + 8 * n * 2 * k * 16 // 25
    out["sequence"] = "crc_square_pool_relin_forms: Square + pooling with one key switch per pooled ciphertext; figures are per SQUARED ciphertext"
out["total"] = dict(read=round(tot_r), write=round(tot_w), sum=round(tot_r + tot_w), algorithmic=alg, ratio=round((tot_r + tot_w) / alg, 2))
print(json.dumps(out, indent=1))
