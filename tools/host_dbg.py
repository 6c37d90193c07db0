"""run crcnn_amd/lib/bench_host on freshly encrypted images with allocation tracing: host_dbg.py [model n k t chunk batch]"""
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
import crcnn_amd as ca
from crcnn_amd.synth import normalize, synth_image
model = sys.argv[1] if len(sys.argv) > 1 else "PlainModelTiny"
n, k, t = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4096, 2, 1 << 32)
chunk, batch = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (128, 256)
group = sys.argv[7] if len(sys.argv) > 7 else "1"
q = ca.default_coeff_modulus_128(n)[:k]
E = ca.Engine(n, q, t, device=-1)
sk, pk = E.keygen(2024)
xs = []
for i in range(2):
    pl, _ = E.encode(normalize(synth_image(i)).reshape(-1))
    xs.append(E.encrypt(pk, pl, 7000 + 1000 * i))
np.ascontiguousarray(np.stack(xs)).tofile("/dev/shm/in.u64")
env = dict(os.environ, CRC_HOST_TRACE="1")
p = subprocess.run(["crcnn_amd/lib/bench_host", model, f"tests/golden/models/{model}.h5", str(n), str(k), str(t), "/dev/shm/in.u64", "2", str(batch), str(chunk), "1", "/dev/shm/out0.u64", group],
                   capture_output=True, text=True, env=env)
print(p.stdout[-2000:]); print(p.stderr[-2500:])
