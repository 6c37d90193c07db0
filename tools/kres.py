#!/usr/bin/env python3
"""Register / scratch / occupancy of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
usage: tools/kres.py crcnn_amd/csrc/kernels_relin64.hip [substring filter ...]"""
import re, subprocess, sys, os
src = sys.argv[1]; filt = sys.argv[2:]
d = os.path.dirname(os.path.abspath(src))
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("KRES_FLAGS","").split() + ["-c", src, "-o", "/dev/null"],
                   cwd=os.getcwd(), capture_output=True, text=True)
cur = None; info = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m: cur = m.group(1); info[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur: info[cur][m.group(1).strip()] = m.group(2)
if r.returncode: print(r.stderr[-3000:]); sys.exit(1)
names = subprocess.run(["c++filt"], input="\n".join(info), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(info, names):
    name = re.sub(r"\(.*", "", name)
    if filt and not any(f in name for f in filt): continue
    v = info[mangled]
    print(f'{name[:84]:84s} vgpr {v.get("VGPRs","?"):>3s} agpr {v.get("AGPRs","?"):>3s} sgpr {v.get("TotalSGPRs","?"):>3s} scratch {v.get("ScratchSize","?"):>4s} occ {v.get("Occupancy","?")} spill {v.get("VGPRs Spill","?")}')
