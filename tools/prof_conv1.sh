cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c1 -o c1 -- python3 tools/check_conv1.py ${1:-4096} ${2:-2} ${3:-32} ${4:-tiny} > gpurun_out/cc1.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_c1/c1_kernel_trace.csv')):
    n=r['Kernel_Name']
    if 'conv1' in n or 'rows1' in n or 'slotmajor' in n or 'pack_tensor' in n or 'mac3' in n:
        print(n[:34], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, 'ms', r['Workgroup_Size_X'], r['Grid_Size_X'], 'vgpr', r['VGPR_Count'], r['Accum_VGPR_Count'], 'scr', r['Scratch_Size'])
PY
