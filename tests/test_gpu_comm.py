"""GPU tests of the multi-GPU start-up path: the engine's RCCL communicator behind the C ABI (crc_comm_*,
crc_broadcast_weights, crc_checksum64) on the one GPU of the test box, and bench.py's multi-rank path end to end (two ranks sharing
the device over the gloo rehearsal backend -- RCCL itself refuses two ranks on one GPU; the 8-GPU run is the driver's)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK = (1 << 64) - 1


def test_comm_one_rank_broadcast_and_checksum():
    import crcnn_amd as ca
    E = ca.Engine(1024, [0x7fffffff380001, 0x3fffffff000001], 1 << 20, device=0)
    rng = np.random.default_rng(7)
    for words in (1, 63, 4096, (1 << 20) + 17):
        w = rng.integers(0, 1 << 63, size=words, dtype=np.uint64)
        d = E.upload(w)
        x, s = E.checksum64(d, words * 8)
        wx = int(np.bitwise_xor.reduce(w))
        ws = int(np.sum(w * (2 * np.arange(words, dtype=np.uint64) + 1), dtype=np.uint64))      # wraps mod 2^64
        assert (x, s) == (wx, ws)
    comm = E.comm_create(1, 0, E.comm_unique_id())
    assert E.L.crc_comm_world(comm) == 1 and E.L.crc_comm_rank(comm) == 0
    E.broadcast_weights(comm, d, words * 8, root=0)
    E.sync()
    assert np.array_equal(E.download(d, (words,)), w)
    got = E.allgather_u64(comm, [x, s, 5])
    assert got.shape == (1, 3) and [int(v) for v in got[0]] == [x, s, 5]
    with pytest.raises(ca.CrcError):
        E.broadcast_weights(comm, d, words * 8, root=1)
    E.comm_destroy(comm)
    E.close()


def test_comm_one_process_all_devices():
    """the one-process, many-GPU entry points (crc_comm_create_all / crc_broadcast_weights_all): one context per visible GPU (one on the test box, the same
    code with 8 on a node), broadcast from the last device, every context's checksum equals the root's, and the calling thread's device is left alone"""
    import ctypes
    import torch
    import crcnn_amd as ca
    ndev = min(torch.cuda.device_count(), 8)
    assert ndev >= 1
    engs = [ca.Engine(1024, [0x7fffffff380001, 0x3fffffff000001], 1 << 20, device=d) for d in range(ndev)]
    L = engs[0].L
    VP = ctypes.c_void_p
    ctxs = (VP * ndev)(*[e.c for e in engs]); comms = (VP * ndev)()
    torch.cuda.set_device(0)
    assert L.crc_comm_create_all(ctxs, ndev, comms) == 0
    assert torch.cuda.current_device() == 0
    words = (1 << 18) + 5
    rng = np.random.default_rng(11)
    root = ndev - 1
    bufs = []
    for d, e in enumerate(engs):
        w = rng.integers(0, 1 << 63, size=words, dtype=np.uint64)
        if d == root:
            want = w
        bufs.append(e.upload(w))
    ptrs = (VP * ndev)(*[engs[d].p(bufs[d]) for d in range(ndev)])
    assert L.crc_broadcast_weights_all(comms, ndev, ctypes.cast(ptrs, ctypes.POINTER(VP)), words, root, None) == 0
    assert torch.cuda.current_device() == 0
    wx = int(np.bitwise_xor.reduce(want)); ws = int(np.sum(want * (2 * np.arange(words, dtype=np.uint64) + 1), dtype=np.uint64))
    for d in reversed(range(ndev)):                  # checksum of context d from a thread whose current device is 0
        engs[d].sync()
        assert engs[d].checksum64(bufs[d], words * 8) == (wx, ws), d
        assert np.array_equal(engs[d].download(bufs[d], (words,)), want)
    for d in range(ndev):
        L.crc_comm_destroy(comms[d])
    assert torch.cuda.current_device() == 0
    for e in engs:
        e.close()


def _run_bench(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny1024", "--steps", "1", "--cpu-seconds", "0", "--also", "none"] + extra,
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_small_workload_matches_reference_golden():
    """bench.py's measured path (N = 1: crcnn_amd/lib/bench_host, the C++ host classes) on the small workload: the output ciphertexts of image 0 are, bit for bit, what the
    compiled reference computed from the same encrypted input (tests/golden/net_tiny1024_eng.json); the line carries the contract fields, the per-layer times from HIP
    events, a roofline entry and the rank's HBM plan.  With --python-twin the same workload through netrun.py, with the reference's layer structure run unfused as well"""
    line = _run_bench(["--gpus", "1", "--python-twin", "--unfused-images", "24"])
    c = line["check"]
    assert line["n_gpus"] == 1 and c["golden_match"] is True and c["golden"] == "net_tiny1024_eng.json" and c["all_ok"] is True
    assert c["predictions_match_plain_model"] in ("4/4", "24/24") and c["last_timed_launch_identical_to_first"] is True
    assert line["config"]["host"].startswith("C++ host classes") and line["value"] > 0 and line["dtype"] == "u64" and line["vs_baseline"] is None
    assert set(line["ms_per_layer"]) == {"pool1_features.conv1+pool1", "pool2_features.conv2+pool2", "classifier.fc3", "classifier.fc4"}
    assert line["roofline"]["frac"] > 0 and line["roofline"]["launch_ms"] > 0 and line["roofline"]["traffic_source"] is None
    assert line["hbm_plan"]["parameters"] > 0 and line["hbm_plan"]["total_bytes"] > line["hbm_plan"]["parameters"]
    tw = line["python_twin"]
    assert tw["check"]["golden_match"] is True and tw["check"]["tiled_outputs_identical"] and tw["reference_layer_structure"]["outputs_identical_to_fused"] is True
    assert 0.5 < tw["vs_host"] < 2.0


def test_bench_two_ranks_end_to_end_on_one_device():
    """the Python twin (--python-twin) at N = 2, self-launched: weights built on rank 0 only, broadcast, per-rank checksums, every rank verified"""
    line = _run_bench(["--gpus", "2", "--python-twin", "--unfused-images", "0"], {"CRC_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    c = line["check"]
    assert c["ranks_verified"] == "2/2" and c["golden_match"] is True and c["all_ok"] is True
    b = line["weight_broadcast"]
    assert b["checksums_match"] == "2/2" and b["bytes"] > 0 and b["seconds"] > 0 and b["via"].startswith("torch.distributed (gloo)")


def _assert_cpp_ranks(line, world):
    assert line["n_gpus"] == world and line["scaling"] == "weak"
    assert line["config"]["host"].startswith("C++ host classes") and line["config"]["parallelism"] == f"image-sharded x{world}"
    c = line["check"]
    assert c["ranks_verified"] == f"{world}/{world}" and c["golden_match"] is True and c["all_ok"] is True
    b = line["weight_broadcast"]
    assert b["via"].startswith("Network::broadcastParameters") and b["bytes"] > 0 and b["seconds"] > 0
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == list(range(world)) and all(r["images"] > 0 and r["elapsed_s"] > 0 and r["parameters"] > 0 for r in pr)
    # whole-job rate = all ranks' images / the slowest rank's time
    assert abs(line["value"] - sum(r["images"] for r in pr) / max(r["elapsed_s"] for r in pr)) < 1e-3 * line["value"]


def test_bench_two_cpp_ranks_rehearsed_on_one_device():
    """`python bench.py --gpus 2` as the driver launches it, with the C++ host as the measured path on EVERY rank: two bench_host children (started by two Python ranks
    that never touch the GPU), rank 0's rendezvous id through the file, crc_comm_create, Network::broadcastParameters, all-gather barriers around the timed steps, the
    elapsed times gathered.  On this one-GPU box the communicator's bytes travel through the shared-memory rehearsal transport (CRC_COMM_TRANSPORT=shm: RCCL refuses two
    ranks on one device); everything above crc_comm_* is the code that runs on eight GPUs"""
    line = _run_bench(["--gpus", "2"], {"CRC_COMM_TRANSPORT": "shm"})
    _assert_cpp_ranks(line, 2)
    assert "shared-memory rehearsal transport" in line["weight_broadcast"]["via"]


def test_bench_streamed_inputs_beside_the_resident_line():
    """--stream-inputs both: after the resident measurement the same network is fed launch by launch over PCIe -- ciphertexts from page-locked memory (same bits out as the
    resident launch), or pixel plaintexts that are encrypted on the device in front of the first layer (fresh ciphertexts: checked by decrypting the logits)"""
    line = _run_bench(["--gpus", "1", "--stream-inputs", "both", "--stream-steps", "2"])
    assert line["check"]["all_ok"] is True
    st = {m["mode"]: m for m in line["streamed"]}
    assert set(st) == {"ciphertext", "plaintext"}
    assert st["ciphertext"]["outputs_identical_to_resident"] is True and st["ciphertext"]["h2d_GBps"] > 0 and st["ciphertext"]["images_per_s"] > 0
    assert st["plaintext"]["check"]["predictions_match_plain_model"].split("/")[0] == st["plaintext"]["check"]["predictions_match_plain_model"].split("/")[1]
    assert st["plaintext"]["bytes_per_image"] * 2 * 2 == st["ciphertext"]["bytes_per_image"]          # a ciphertext is 2 polys x k = 2 residues of its plaintext's size


def test_bench_two_ranks_over_rccl():
    """the multi-GPU path as the driver launches it, on real RCCL: two C++ bench_host ranks on two GPUs (`python bench.py --gpus 2`, self-launch), the encoded weights
    built on rank 0 only and sent by Network::broadcastParameters (crc_broadcast_weights: ncclBroadcast), every rank's device checksum equal to the root's, every rank's
    output ciphertexts verified against the reference golden.  Needs two visible GPUs: skipped on the one-GPU boxes this repository is developed on (the path is rehearsed there
    over gloo, test_bench_two_ranks_end_to_end_on_one_device)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least two GPUs")
    line = _run_bench(["--gpus", "2"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    _assert_cpp_ranks(line, 2)
    assert "ncclBroadcast" in line["weight_broadcast"]["via"]
    assert line["hbm_plan"]["parameters"] > 0


def test_bench_two_cpp_ranks_encode_their_own_weights():
    """--weights-via floats (SURVEY 8e's alternative to the residue broadcast): every rank lifts + transforms the model's weights itself, Network::broadcastParameters sends
    only the evaluation keys -- and still compares every rank's parameter checksum with the root's, so two ranks that encoded differently could not pass"""
    line = _run_bench(["--gpus", "2", "--weights-via", "floats"], {"CRC_COMM_TRANSPORT": "shm"})
    _assert_cpp_ranks(line, 2)
    b = line["weight_broadcast"]
    assert b["weights_via"].startswith("floats")
    other = _run_bench(["--gpus", "2"], {"CRC_COMM_TRANSPORT": "shm"})
    assert other["weight_broadcast"]["weights_via"].startswith("broadcast") and other["weight_broadcast"]["bytes"] > 20 * b["bytes"]     # only the keys travelled


def test_bench_a_failing_rank_stops_the_job_with_a_readable_line():
    """crc_comm_create failing on one rank (here: a rendezvous id that names a shared-memory segment nobody created) must surface as a non-zero exit of that rank's
    bench_host with a line a person can read, and the sibling rank must stop too instead of waiting in a collective (never a re-exec, never a hang)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"CRC_COMM_TRANSPORT": "shm", "CRC_TEST_BREAK_COMM_RANK": "1"})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny1024", "--steps", "1", "--cpu-seconds", "0", "--also", "none", "--gpus", "2"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode != 0
    assert "crc_comm_create failed" in out.stderr and "bench_host failed on rank 1" in out.stderr, out.stderr[-2000:]


def test_bench_latency_published_and_valu_lines():
    """the new lines of the default invocation on the small ring: single-image latency (batch 1, a synchronisation per image), the no-matrix-core pass (every conv / dense
    layer on mac3_kernel) with the same golden ciphertexts"""
    line = _run_bench(["--gpus", "1", "--latency", "on"])
    assert line["check"]["all_ok"] is True
    lat = line["latency"][0]
    assert lat["latency_ms"] > 0 and lat["check"]["golden_match"] is True and set(lat["ms_per_layer"]) == set(line["ms_per_layer"]) and lat["images_timed"] >= 5
    assert abs(sum(lat["ms_per_layer"].values()) - lat["latency_ms"]) < 0.5 * lat["latency_ms"]
    assert line["config"]["latency_ms_single_image"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for cfg, extra in (("tiny2048r", []), ("approx4096r", ["--batch", "128"]), ("tiny4096_valu", ["--batch", "64", "--distinct", "2"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "1", "--cpu-seconds", "0", "--also", "none", "--gpus", "1"] + extra,
                             capture_output=True, text=True, env=env, timeout=900)
        assert out.returncode == 0, (cfg, out.stdout[-1500:], out.stderr[-3000:])
        ln = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        c = ln["check"]
        assert c["golden_match"] is True and c["all_ok"] is True and c["last_timed_launch_identical_to_first"] is True, (cfg, c)
        if cfg.endswith("r"):
            vp = ln["vs_published"]
            assert "T_REENC" in ln["ms_per_layer"] and ln["ms_per_layer"]["T_REENC"] > 0 and vp["vs_published"] > 100 and 0 < vp["refresh_share_of_image"] < 0.5
            assert [r_["columns"] for r_ in vp["per_column"]].count("T_REENC") == 1
        else:
            assert all(k.startswith("mac3_kernel") or k.startswith("mac") for k in ln["mac_kernel_per_layer"].values()), ln["mac_kernel_per_layer"]
            assert "mfma" not in json.dumps(ln["mac_kernel_per_layer"])


def test_bench_four_cpp_ranks_rehearsed_on_one_device():
    """the same rehearsal with FOUR ranks (the box allows six GPU processes): the all-gathered placement / checksum / timing arrays, the rendezvous directory and the
    failure markers are sized by the world, and a two-rank run exercises none of the indexing beyond rank 1.  Both weight-distribution strategies."""
    for extra in ([], ["--weights-via", "floats"]):
        line = _run_bench(["--gpus", "4"] + extra, {"CRC_COMM_TRANSPORT": "shm"})
        _assert_cpp_ranks(line, 4)
        assert line["config"]["parallelism"] == "image-sharded x4"
