cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G=${1:-conv2p}
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
tag=$(echo $c | cut -d' ' -f1)
rm -rf gpurun_out/pmcm_$tag
timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcm_$tag -o p -- python3 tools/bench_mac.py $G 32 2 limbk > gpurun_out/pmcm_$tag.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmcm_$tag/p_counter_collection.csv')):
    if 'mfma_mac' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, [round(x/1e6,1) for x in v][-2:])
dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open('gpurun_out/pmcm_$tag/p_kernel_trace.csv')) if 'mfma_mac' in r['Kernel_Name']]
print('durations ms', dur[-2:])
PY
done
