# issue / wait / LDS counters per kernel NAME of any command (two rocprofv3 --pmc passes, counters only: no other trace domain):
#   tools/pmc_issue.sh <tag> <program and arguments ...>      ->  gpurun_out/pmc_issue_<tag>.json
# share = part of the SIMD cycles the kernel had (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): VALU issue at 4 cycles per wave instruction (the fp64 rate, and
# what one wave alone gets for the 32-bit ones), LDS = the LDS pipe's busy cycles per CU, conflicts = the extra cycles bank conflicts add to them.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/pmc_issue_$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/a -o a -- "$@" > $O/a.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/b -o b -- "$@" > $O/b.log 2>&1 &&
python3 - "$O" "$TAG" "$*" <<'PY'
import csv, collections, glob, json, sys
O, tag, cmd = sys.argv[1:4]
def load(path):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); t = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"]); t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return acc, t, n
a, ta, na = load(O + "/a"); b, tb, nb = load(O + "/b")
out = {}
for k in sorted(a, key=lambda k: -ta[k]):
    if k not in b or ta[k] < 0.05: continue
    A, B = a[k], b[k]
    cyc = B["GRBM_GUI_ACTIVE"] / 8
    simd = cyc * 1024
    if simd <= 0: continue
    wc = A["SQ_WAVE_CYCLES"] or 1.0
    out[k] = dict(launches=len(na[k]), ms_total_under_pmc=round(ta[k], 3), effective_clock_GHz=round(cyc / (tb[k] * 1e-3) / 1e9, 3),
                  valu_instructions=A["SQ_INSTS_VALU"], lds_instructions=B["SQ_INSTS_LDS"],
                  vmem_busy_share_of_cu_cycles=round(B["SQ_ACTIVE_INST_VMEM"] * 4 / (cyc * 256), 3),
                  valu_issue_share_at_4_cycles=round(A["SQ_INSTS_VALU"] * 4 / simd, 3),
                  lds_busy_share_of_cu_cycles=round(B["SQ_ACTIVE_INST_LDS"] * 4 / (cyc * 256), 3),
                  lds_bank_conflict_share_of_cu_cycles=round(B["SQ_LDS_BANK_CONFLICT"] / (cyc * 256), 3),
                  wave_cycles_waiting=round(A["SQ_WAIT_ANY"] / wc, 3), wave_cycles_issue_stalled=round(A["SQ_WAIT_INST_ANY"] / wc, 3),
                  wave_cycles_issuing=round(A["SQ_ACTIVE_INST_ANY"] / wc, 3), wave_cycles_lds_issue_stalled=round(B["SQ_WAIT_INST_LDS"] / wc, 3),
                  mean_waves_per_simd=round(wc * 4 / simd, 2))
json.dump(dict(command=cmd, note="two rocprofv3 --pmc passes, summed per kernel name over all its launches; SQ_* wave-cycle counters are in quad-cycles",
               kernels=out), open(f"gpurun_out/pmc_issue_{tag}.json", "w"), indent=1)
for k, v in out.items():
    print(k[:70], {x: v[x] for x in ("ms_total_under_pmc", "valu_issue_share_at_4_cycles", "lds_busy_share_of_cu_cycles", "lds_bank_conflict_share_of_cu_cycles",
                                        "wave_cycles_waiting", "wave_cycles_issue_stalled", "mean_waves_per_simd", "effective_clock_GHz")})
PY
