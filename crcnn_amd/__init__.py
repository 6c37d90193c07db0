"""crcnn_amd -- MI355X-native encrypted-CNN evaluation engine (drop-in for CrCNN's evaluation path).

The product is libcrcnn_hip.so (C ABI: include/crcnn_hip.h; kernels: crcnn_amd/csrc/*.hip) plus the C++ host classes
in crcnn_amd/host/ that mirror CrCNN's Layer / Network / CnnBuilder interface.  This Python package is only the
ctypes plumbing used by the tests and bench.py.  Importing it loads the HIP library and fails if it is missing.
"""
from . import binding
from .binding import COEFF, NTT, NTTP, NTTL, NTTL1, NTTLC, CrcError, Engine, default_coeff_modulus_128, h5_list, h5_read, load  # noqa: F401

load()
