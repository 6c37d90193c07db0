"""world_size-2 test of the N>1 plumbing on CPU (gloo): shard ranges partition the batch, the weight broadcast delivers
rank 0's buffers bit-exactly in chunks, timing is the max over ranks.  (The kernels themselves need a GPU; what is multi-
process about the path is exactly this plumbing -- there is no data-path collective.)"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from crcnn_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weights: rank 0 holds the encoded parameters, the others allocate empty buffers of the same shapes
        g = torch.Generator().manual_seed(1234)
        ref = [torch.randint(0, 1 << 62, (n,), dtype=torch.int64, generator=g) for n in (1000, 70001, 3)]
        bufs = [r.clone() if rank == 0 else torch.zeros_like(r) for r in ref]
        sent = shard.broadcast_buffers(bufs, src=0, chunk_bytes=64 * 1024)        # forces multi-piece broadcasts
        ok_bcast = all(torch.equal(a, b) for a, b in zip(bufs, ref)) and sent == sum(r.numel() * 8 for r in ref)
        # image sharding
        b, e = shard.shard_range(1027, rank, world)
        total = shard.gather_counts(e - b, torch.device("cpu"))
        # timing = max over ranks
        tmax = shard.max_over_ranks(1.0 + rank, torch.device("cpu"))
        dist.barrier()
        out.put((rank, ok_bcast, (b, e), total, tmax))
    finally:
        dist.destroy_process_group()


def test_two_process_plumbing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == (0, 514) and res[1][2] == (514, 1027)
    assert all(r[3] == 1027 for r in res)
    assert all(r[4] == 2.0 for r in res)


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from crcnn_amd import shard
    for total in (0, 1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            ranges = [shard.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE must start its own two ranks (fresh children of a parent that has
    made no GPU call), relay rank 0's JSON line and propagate the exit code.  --launch-check stops after the rendezvous (no GPU here)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert (line["launch_check"], line["n_gpus"], line["ranks_seen"], line["self_launched"]) == (True, 2, 2, True)
    # the launcher of the measured path: one C++ bench_host child per rank, rank 0's rendezvous id reached the other through the file, and the ranks split the host cores
    assert line["host"].startswith("C++ bench_host") and line["same_rendezvous_id_on_every_rank"] is True
    assert line["host_threads_per_rank"] == [max(1, min(16, (os.cpu_count() or 1) // 2))]
    # a rank that fails makes the parent fail: --gpus 3 announced, but the check insists on what the flag says
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], capture_output=True, text=True, env=env, timeout=300)
    assert bad.returncode != 0


def test_bench_launches_eight_ranks():
    """the first 8-GPU contact of the launcher, rehearsed on CPU: `python bench.py --gpus 8 --launch-check` starts eight harness ranks, each starts its bench_host child,
    rank 0's rendezvous id reaches the other seven through the file in the 0700 directory rank 0 made, and the ranks split the host cores eight ways"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert (line["launch_check"], line["n_gpus"], line["ranks_seen"], line["self_launched"]) == (True, 8, 8, True)
    assert line["same_rendezvous_id_on_every_rank"] is True
    assert line["host_threads_per_rank"] == [max(1, min(16, (os.cpu_count() or 1) // 8))]
