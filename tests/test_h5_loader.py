"""The built-in HDF5 reader (crcnn_amd/csrc/h5lite.cpp) against libhdf5's own view of the reference's model files.

tests/golden/h5_datasets.npz was produced in the build container with the image's `h5dump -b LE` (oracle/make_golden.py
loader); tests/golden/models/*.h5 are the reference's data artefacts (PlainModel/*.h5), needed to run its configs."""
import hashlib
import os

import numpy as np
import pytest

import crcnn_amd as ca

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODELS = ["PlainModelTiny", "ApproxPlainModel", "PlainModelWoPad"]


@pytest.mark.parametrize("model", MODELS)
def test_every_dataset_bit_exact(model):
    g = np.load(os.path.join(GOLD, "h5_datasets.npz"))
    path = os.path.join(GOLD, "models", model + ".h5")
    names = sorted({k.split("/")[1] for k in g.files if k.startswith(model + "/")})
    assert sorted(ca.h5_list(path)) == names
    for nm in names:
        if nm.endswith("num_batches_tracked"):       # int64 scalar bookkeeping entry of torch BatchNorm: not a float dataset
            with pytest.raises(ca.CrcError):
                ca.h5_read(path, nm)
            continue
        a = ca.h5_read(path, nm)
        assert a.dtype == np.float32 and a.size == int(g[f"{model}/{nm}/count"])
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g[f"{model}/{nm}/sha256"]), nm
        assert np.array_equal(a[:8], g[f"{model}/{nm}/head"][: a.size])


def test_known_first_weight():
    # conv1.weight of PlainModelTiny has 800 floats, w[0] = 0.020209
    a = ca.h5_read(os.path.join(GOLD, "models", "PlainModelTiny.h5"), "pool1_features.conv1.weight")
    assert a.size == 800 and abs(float(a[0]) - 0.020209) < 1e-6


def test_errors():
    path = os.path.join(GOLD, "models", "PlainModelTiny.h5")
    with pytest.raises(ca.CrcError):
        ca.h5_read(path, "no.such.dataset")
    with pytest.raises(ca.CrcError):
        ca.h5_read(os.path.join(GOLD, "h5_datasets.npz"), "x")      # not an HDF5 file
    with pytest.raises(ca.CrcError):
        ca.h5_read("/nonexistent/file.h5", "x")


def test_malformed_files_fail_cleanly():
    """message bodies, dimension products and base-relative addresses are bounds-checked -- truncated or mutated files give CRC_ERR_IO / not-found /
    a clean read, never an out-of-bounds access.  Run in a child process so that a crash cannot take the test session down."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import crcnn_amd as ca
src = open(%r, "rb").read()
rng = np.random.default_rng(3)
tmp = %r
names = ["pool1_features.conv1.weight", "classifier.fc3.bias"]
bad = 0
for it in range(300):
    b = bytearray(src)
    if it %% 3 == 0:
        b = b[: int(rng.integers(8, len(b)))]                       # truncation
    else:
        for _ in range(int(rng.integers(1, 6))):                      # header mutations: superblock, B-tree, heap, object headers live in the first pages
            pos = int(rng.integers(0, min(len(b), 8192)))
            b[pos] = int(rng.integers(0, 256)) if it %% 3 == 1 else 0xff
    open(tmp, "wb").write(bytes(b))
    for nm in names:
        try:
            ca.h5_read(tmp, nm)
        except ca.CrcError:
            bad += 1
    try:
        ca.h5_list(tmp)
    except ca.CrcError:
        bad += 1
print("survived", bad)
''' % (os.path.dirname(GOLD.rstrip("/")) + "/..", os.path.join(GOLD, "models", "PlainModelTiny.h5"), os.path.join(os.environ.get("TMPDIR", "/tmp"), "crc_fuzz.h5"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "survived" in out.stdout, (out.returncode, out.stderr[-1500:])
    assert int(out.stdout.split()[-1]) > 50          # most mutations must have been detected as errors
