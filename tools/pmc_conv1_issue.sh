# issue / wait / LDS-conflict counters of mfma_conv1_kernel (two rocprofv3 --pmc passes over tools/check_conv1.py 4096 2 128 tiny); writes gpurun_out/pmc_conv1_issue.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_conv1_issue; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/a -o a -- python3 tools/check_conv1.py 4096 2 128 tiny > $O/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/b -o b -- python3 tools/check_conv1.py 4096 2 128 tiny > $O/b.log 2>&1
python3 - <<PY
import csv, collections, glob, json
def last(path):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); t = {}
    for r in csv.DictReader(open(f)):
        if "mfma_conv1_kernel" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    k = sorted(acc, key=int)[-1]          # the last launch: the limb-tensor output of 128 images
    return dict(acc[k]), t[k]
a, ms = last("$O/a"); b, _ = last("$O/b")
cyc = b["GRBM_GUI_ACTIVE"] / 8          # (summed over the 8 XCDs)
simd = cyc * 256 * 4
d = dict(effective_clock_GHz=round(cyc / (ms * 1e-3) / 1e9, 3), simd_cycles_available=simd, valu_instructions_per_mfma=round(a["SQ_INSTS_VALU"] / b["SQ_INSTS_MFMA"], 2),
         valu_instructions_per_output=round(a["SQ_INSTS_VALU"] * 64 / (128 * 8192 * 9216), 1),
         valu_issue_share_at_4_cycles_per_instruction=round(a["SQ_INSTS_VALU"] * 4 / simd, 3), mfma_pipe_busy_share=round(b["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 3),
         lds_bank_conflict_share_of_cu_cycles=round(b["SQ_LDS_BANK_CONFLICT"] / (cyc * 256), 3))
json.dump(dict(kernel="mfma_conv1_kernel (tools/check_conv1.py 4096 2 128 tiny: conv1+pool1 of PlainModelTiny, 128 images per launch, limb-tensor output), two rocprofv3 --pmc passes",
               launch_ms_under_pmc=round(ms, 2), pass_a=a, pass_b=b, derived=d), open("gpurun_out/pmc_conv1_issue.json", "w"), indent=1)
print(json.dumps(d), ms)
PY
