"""Whole networks, oracle vs the reference: per-layer SHA-256 of the real models (real weights, one synthetic encrypted
image) at n=256 must equal what the compiled CrCNN Network::forward produced (tests/golden/net_*256.json)."""
import numpy as np
import pytest

from netcommon import load_net_golden, make_inputs, model_weights, oracle_forward, sha


@pytest.mark.parametrize("name", ["tiny256", "approx256", "wopad256"])
def test_oracle_network_digests(name):
    g = load_net_golden(name)
    O, sk, pk, evk, img, x = make_inputs(g)
    W = model_weights(g["model"])
    last = None
    for i, t in oracle_forward(O, g["model"], W, x, evk):
        assert sha(t) == g["layers"][i]["sha256"], (name, i, g["layers"][i]["name"])
        last = t
    assert sha(last) == g["out_sha256"]
    assert [O.noise_budget(sk, last[0, j, 0]) for j in range(3)] == g["budget"][:3]
