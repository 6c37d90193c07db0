"""Synthetic workload inputs (the MNIST test images are not part of the reference checkout: .MISSING_LARGE_BLOBS).

Seeded MNIST-like 28x28 uint8 images (81 % background zeros, the rest uniform in 1..255) and the reference's
float32 normalisation (CrCNN/src/utils.cpp:9-18,27)."""
import numpy as np

_M = (1 << 64) - 1


def synth_image(index, seed=0xC0FFEE):
    s = (seed + index) & _M
    out = np.zeros(784, dtype=np.uint8)
    for i in range(784):
        s = (s + 0x9E3779B97F4A7C15) & _M
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
        z ^= z >> 31
        if (z & 0xFFFF) >= int(0.81 * 65536):
            out[i] = 1 + ((z >> 16) % 255)
    return out.reshape(28, 28)


def normalize(img_u8):
    p = img_u8.astype(np.float32)
    return ((p / np.float32(255)) - np.float32(0.1307)) / np.float32(0.3081)
