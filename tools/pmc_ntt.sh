# PMC view of the row NTT (separate passes per counter group; no trace domains combined with --pmc)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_ntt; mkdir -p $O
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -o p -- python3 tools/bench_ntt.py 8192 3 4096 > $O/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ntt/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ntt_rows_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:26s} avg/launch {sum(v)/len(v):16.1f}  launches {len(v)}")
PY
