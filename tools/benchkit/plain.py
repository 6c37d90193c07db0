"""float64 forward of the plaintext network: what the decrypted logits of bench.py's outputs are compared with"""
import numpy as np


def plain_forward(model, W, img):
    """float64 numpy forward of the plaintext network (PlainModel/*.py semantics) for the prediction check"""
    from crcnn_amd.netrun import TOPOLOGIES
    x = img.astype(np.float64)[None]
    for kind, name, a in TOPOLOGIES[model]:
        if kind == "conv":
            w = W[name + ".weight"].astype(np.float64).reshape(a["nf"], a["zd"], a["xf"], a["yf"]); b = W[name + ".bias"].astype(np.float64)
            xo, yo = (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1
            y = np.zeros((a["nf"], xo, yo))
            for i in range(xo):
                for j in range(yo):
                    patch = x[:, i * a["xs"]:i * a["xs"] + a["xf"], j * a["ys"]:j * a["ys"] + a["yf"]]
                    y[:, i, j] = (w * patch[None]).sum(axis=(1, 2, 3)) + b
            x = y
        elif kind in ("pool", "avgpool"):
            xo, yo = (a["xd"] - a["xf"]) // a["xs"] + 1, (a["yd"] - a["yf"]) // a["ys"] + 1
            y = np.zeros((a["zd"], xo, yo))
            for i in range(xo):
                for j in range(yo):
                    y[:, i, j] = x[:, i * a["xs"]:i * a["xs"] + a["xf"], j * a["ys"]:j * a["ys"] + a["yf"]].sum(axis=(1, 2))
            x = y / (a["xf"] * a["yf"]) if kind == "avgpool" else y
        elif kind == "bn":
            mean = W[name + ".running_mean"].astype(np.float64); var = W[name + ".running_var"].astype(np.float64)
            x = (x - mean[:, None, None]) / np.sqrt(var + 1e-5)[:, None, None]
        elif kind == "square":
            x = x * x
        elif kind == "fc":
            w = W[name + ".weight"].astype(np.float64).reshape(a["out_dim"], a["in_dim"]); b = W[name + ".bias"].astype(np.float64)
            x = (w @ x.reshape(-1) + b).reshape(1, a["out_dim"], 1)
    return x.reshape(-1)
