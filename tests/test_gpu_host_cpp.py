"""GPU parity of the C++ host classes (crcnn_amd/host: Layer / Network / CnnBuilder with the reference's signatures):
the driver binary crcnn_amd/lib/test_host builds the real models with CnnBuilder from the HDF5 files and runs
Network::forward; per-layer outputs must hash to what the compiled reference produced (tests/golden/net_*.json)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from netcommon import GOLD, load_net_golden, make_inputs, sha

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "crcnn_amd", "lib", "test_host")


def run_driver(name, resident, batch=1):
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    d = tempfile.mkdtemp()
    np.array([g["n"], len(g["q"]), g["t"]] + g["q"], dtype=np.uint64).tofile(os.path.join(d, "params.u64"))
    evk.tofile(os.path.join(d, "evk.u64")); x.tofile(os.path.join(d, "net_in.u64"))
    h5 = os.path.join(GOLD, "models", g["model"] + ".h5")
    subprocess.check_call([DRIVER, "net", g["model"], h5, d, "1" if resident else "0", str(batch)])
    return g, O, d


@pytest.mark.parametrize("name", ["tiny256", "approx256", "wopad256"])
def test_cpp_network_layerwise_digests(name):
    g, O, d = run_driver(name, resident=False)
    for i, L in enumerate(g["layers"]):
        t = np.fromfile(os.path.join(d, f"layer_{i}.u64"), dtype=np.uint64)
        assert sha(t) == L["sha256"], (name, i, L["name"])


@pytest.mark.parametrize("name", ["tiny256", "approx256", "wopad256"])
def test_cpp_network_resident_batch(name):
    g, O, d = run_driver(name, resident=True, batch=2)
    out = np.fromfile(os.path.join(d, "out.u64"), dtype=np.uint64).reshape(2, -1)
    assert sha(out[0]) == g["out_sha256"] and sha(out[1]) == g["out_sha256"]


def test_cpp_api_behaviour():
    """client-side round trip, resident == layerwise, refresh path, SEAL-format save/load, exceptions, HDF5 via CnnBuilder"""
    d = tempfile.mkdtemp()
    h5 = os.path.join(GOLD, "models", "PlainModelTiny.h5")
    out = subprocess.run([DRIVER, "api", h5, d], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "api ok" in out.stdout
