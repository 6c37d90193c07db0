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
    # (the built-in reader is what is fuzzed: files it refuses go to libhdf5 when the machine has it, and that library's behaviour on damaged files is its own)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, CRC_H5_BACKEND="lite"))
    assert out.returncode == 0 and "survived" in out.stdout, (out.returncode, out.stderr[-1500:])
    assert int(out.stdout.split()[-1]) > 50          # most mutations must have been detected as errors


VARIANT = os.path.join(GOLD, "h5_variants", "tiny_subset_v3_chunked_gzip.h5")


def _read_in_child(backend, path, names):
    """h5_read in a fresh process with CRC_H5_BACKEND set (the library reads its environment per call, but a child also keeps a crash inside the HDF5 library out of
    the test session); returns {name: sha256 | error}"""
    import json
    import subprocess
    import sys
    code = (r"import sys, json, hashlib; sys.path.insert(0, %r); import crcnn_amd as ca; out = {'available': bool(ca.binding.load().crc_h5_backend_available())}" "\n"
            r"try: out['list'] = sorted(ca.h5_list(%r))" "\n" r"except Exception as e: out['list'] = 'error: ' + str(e)" "\n"
            r"for nm in %r:" "\n" r"    try: out[nm] = hashlib.sha256(ca.h5_read(%r, nm).tobytes()).hexdigest()" "\n" r"    except Exception as e: out[nm] = 'error: ' + str(e)" "\n"
            r"print(json.dumps(out))") % (os.path.dirname(os.path.dirname(GOLD)), path, list(names), path)
    env = dict(os.environ)
    if backend:
        env["CRC_H5_BACKEND"] = backend
    else:
        env.pop("CRC_H5_BACKEND", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr[-1500:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_other_layouts_go_through_libhdf5():
    """the reference reads its models through libhdf5 and so loads any layout; the built-in reader parses only what PlainModel/ToH5.py writes.  The fixture is
    four datasets of PlainModelTiny.h5 re-written by the image's own h5copy / h5repack (data, not code) with superblock version 3, chunked layouts and a shuffle + gzip
    filter: the built-in reader must refuse it, the dlopen'ed libhdf5 must give the very floats of the original file -- and give them for the original file too."""
    names = ["pool1_features.conv1.weight", "pool1_features.conv1.bias", "classifier.fc4.weight", "classifier.fc4.bias"]
    g = np.load(os.path.join(GOLD, "h5_datasets.npz"))
    want = {nm: str(g[f"PlainModelTiny/{nm}/sha256"]) for nm in names}
    lite = _read_in_child("lite", VARIANT, names)
    assert all(str(lite[nm]).startswith("error") for nm in names) and str(lite["list"]).startswith("error"), "the built-in reader claims a layout it does not parse"
    auto = _read_in_child(None, VARIANT, names)
    if not auto["available"]:
        assert all(str(auto[nm]).startswith("error") for nm in names)          # no libhdf5 on this machine: a loud failure, not a wrong read
        pytest.skip("libhdf5 (>= 1.10) is not installed here: only the refusal could be checked")
    assert auto["list"] == sorted(names)
    assert {nm: auto[nm] for nm in names} == want
    # ... and libhdf5 forced on the reference's own file agrees with the built-in reader's goldens
    forced = _read_in_child("hdf5", os.path.join(GOLD, "models", "PlainModelTiny.h5"), names)
    assert {nm: forced[nm] for nm in names} == want
