for c in 0 2048 4096; do
CRC_SQ_CHUNK=$c python3 bench.py --config approx8192 --steps 2 --cpu-seconds 0 --also none --latency off --stream-inputs none > /tmp/o.json 2>/tmp/o.err || { echo fail $c; tail -3 /tmp/o.err; continue; }
python3 -c "
import json; l=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('CRC_SQ_CHUNK=$c', l['value'], l['ms_per_layer'], l['hbm_plan']['work_buffer']/2**30, l['check']['all_ok'])"
done
