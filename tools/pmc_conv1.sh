cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf gpurun_out/pmc_$c
timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o p -- python3 tools/check_conv1.py 4096 2 32 tiny > gpurun_out/pmc_$c.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc_$c/p_counter_collection.csv')):
    agg[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'conv1' in k or 'rows1' in k or 'slotmajor' in k or 'pack_tensor' in k: print('$c', k, [round(x/1e6,1) for x in v])
PY
done
