cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
for rep in 1 2; do for w in 0 -1; do
  CRC_F64_WAVE=$w timeout -k 10 400 python bench.py --config approx8192 --also none --steps 2 --cpu-seconds 0 --stream-inputs none --distinct 4 > $O/approx_w${w}_$rep.json 2> $O/err.txt
  CRC_F64_WAVE=$w timeout -k 10 400 python bench.py --config wopad16384 --also none --steps 2 --cpu-seconds 0 --stream-inputs none --distinct 4 > $O/wopad_w${w}_$rep.json 2>> $O/err.txt
done; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05o/*_w*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], d["value"], d["check"]["all_ok"], d["ms_per_layer"].get("act1+pool2"))
PY
( time CRC_COMM_TRANSPORT=shm timeout -k 10 600 python bench.py --gpus 2 --steps 2 --also none --cpu-seconds 0 --stream-inputs ciphertext ) > $O/two_ranks_tiny4096_full_size.json 2> $O/two_ranks.err; tail -3 $O/two_ranks.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r05o/two_ranks_tiny4096_full_size.json"):
    if l.startswith("{"):
        d = json.loads(l); print(d["n_gpus"], d["value"], d["check"]["ranks_verified"], d["weight_broadcast"], d["per_rank"], [ (m["mode"], m["images_per_s"]) for m in d["streamed"]])
PY
