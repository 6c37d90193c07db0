"""rank bookkeeping of bench.py: self-launch of the ranks, the CPU-only launch check, and the two kinds of collective the script needs"""
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (torch.distributed.run, one per GPU) as a CHILD process -- this
    parent has made no GPU call and never execs -- relay rank 0's JSON line, and exit non-zero if any rank does."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC (RCCL across processes)
    env["CRC_SELF_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n"); rc = 1
    return rc


def launch_check(args):
    """CPU test of the launcher (tests/test_multiproc_gloo.py): the Python ranks rendezvous over gloo, every rank starts its bench_host child with launch_check=1 -- the
    file rendezvous of the real run with a random id in place of RCCL's, and the per-rank share of the host threads -- and the ranks compare what the children saw.
    Nothing touches a GPU."""
    import subprocess
    ranks = HostRanks(args)
    exe = os.path.join(ROOT, "crcnn_amd", "lib", "bench_host")
    ok, seen, rec = True, ranks.world, None
    if not os.path.exists(exe):
        ok = False
    else:
        p = subprocess.run([exe, "launch_check=1", f"rank={ranks.rank}", f"world={ranks.world}", f"local_world={ranks.local_world}", "rendezvous=" + ranks.rendezvous_path("check")],
                           capture_output=True, text=True, timeout=900)
        if p.returncode == 0:
            rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        else:
            sys.stderr.write(p.stderr[-400:]); ok = False
    recs = ranks.gather(rec)
    ids = {r["id_fnv1a"] for r in recs if r}
    ok = ok and all(recs) and len(ids) == 1 and sorted(r["rank"] for r in recs) == list(range(ranks.world))
    seen = sum(1 for r in recs if r)
    if ranks.rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": ranks.world, "ranks_seen": seen, "self_launched": bool(os.environ.get("CRC_SELF_LAUNCHED")),
                          "host": "C++ bench_host (one child per rank, file rendezvous)", "same_rendezvous_id_on_every_rank": len(ids) == 1,
                          "host_threads_per_rank": sorted({r["host_threads"] for r in recs if r})}), flush=True)
    ranks.close()
    return 0 if ok and seen == args.gpus else 1


def gpu_count_without_hip():
    """GPUs of this node from the KFD topology in sysfs (a node with SIMDs is a GPU): no HIP call, so the harness ranks stay off the devices.  torch.cuda.device_count()
    may initialise the runtime where the amdsmi path is unavailable"""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(ln.split()[:2] for ln in open(f).read().splitlines() if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except OSError:
            pass
    return n


class HostRanks:
    """the Python side of a multi-rank C++ run: this process is ONE rank's harness (torch.distributed.run set RANK / WORLD_SIZE / LOCAL_RANK) and never touches a GPU.
    gloo (CPU) carries three tiny things: a fresh token for the rendezvous file name, "did any child fail", and the count of ranks whose outputs verified."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0")); self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # the rendezvous files (RCCL's 128-byte id, failure markers) live in a directory only this user can enter: rank 0 makes it (mkdtemp: mode 0700, an
        # unpredictable name) and tells the others its path over gloo -- nobody else on the host can pre-create, symlink or read what the ranks exchange there
        import tempfile
        base = "/dev/shm" if os.path.isdir("/dev/shm") else None
        self.dir = tempfile.mkdtemp(prefix="crc_rdv_", dir=base) if self.rank == 0 else None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
            obj = [self.dir]
            dist.broadcast_object_list(obj, src=0)
            self.dir = obj[0]

    def rendezvous_path(self, tag):
        return os.path.join(self.dir, f"rendezvous_{tag}")

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def any(self, flag):
        return any(self.gather(bool(flag)))

    def count(self, flag):
        return sum(int(bool(f)) for f in self.gather(bool(flag)))

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if self.rank == 0:                                           # the rendezvous ids (and failure markers) of this job
            import shutil
            shutil.rmtree(self.dir, ignore_errors=True)


class Dist:
    """rank bookkeeping + the two kinds of collective this script needs: the weight broadcast (RCCL through the engine's own C ABI,
    crc_comm_* / crc_broadcast_weights) and tiny host-side reductions (timing, verification counts)"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0")); self.world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        ndev = torch.cuda.device_count()
        self.backend = os.environ.get("CRC_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm; "gloo" only for single-GPU rehearsals of this code path
        if self.world > 1 and self.backend == "nccl" and ndev < self.world:
            raise SystemExit(f"--gpus {self.world} needs {self.world} GPUs, {ndev} visible (CRC_DIST_BACKEND=gloo rehearses the multi-rank path on fewer)")
        self.local = local % max(1, ndev)                # (rehearsals with more ranks than GPUs share a device; the driver uses one rank per GPU)
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.comm = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev, timeout=datetime.timedelta(minutes=10))      # a collective nobody else joins ends the job, not hangs it
            else:
                dist.init_process_group(self.backend, timeout=datetime.timedelta(minutes=10))

    def make_comm(self, E):
        """RCCL communicator of the engine (C ABI); the 128-byte rendezvous id travels over the torch.distributed store"""
        if self.world == 1 or self.backend != "nccl":
            return None, None
        comm, err = None, None
        # every rank runs the SAME sequence of torch.distributed collectives whatever fails: rank 0 always broadcasts (id or None, error), every rank skips
        # crc_comm_create when there is no id, and the all-reduce afterwards tells everybody whether ALL ranks hold a communicator (otherwise all of them use
        # the torch.distributed broadcast).  A rank that dies inside ncclCommInitRank is caught by the process group's timeout (init_process_group above).
        obj = [None, None]
        if self.rank == 0:
            try:
                obj = [E.comm_unique_id(), None]
            except Exception as ex:
                obj = [None, f"{type(ex).__name__}: {ex}"]
        self.dist.broadcast_object_list(obj, src=0)
        if obj[0] is None:
            err = obj[1] or "rank 0 could not make a rendezvous id"
        else:
            try:
                comm = E.comm_create(self.world, self.rank, obj[0])
            except Exception as ex:
                err = f"{type(ex).__name__}: {ex}"
        have = self.sum(int(comm is not None))
        if have != self.world:
            if comm is not None:
                E.comm_destroy(comm)
            errs = [e_ for e_ in self.gather(err) if e_]
            return None, (errs[0] if errs else "communicator missing on some rank")
        return comm, None

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, v):
        if self.world == 1:
            return v
        from crcnn_amd import shard
        return shard.max_over_ranks(v, self.dev)

    def sum(self, v):
        if self.world == 1:
            return v
        from crcnn_amd import shard
        return shard.gather_counts(v, self.dev)

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out
