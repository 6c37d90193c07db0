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
