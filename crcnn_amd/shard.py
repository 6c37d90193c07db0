"""Multi-GPU plumbing of the evaluation path (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests).  Images are independent, so the path shards with NO data-path collective; the only
collective is the start-up broadcast of the encoded (NTT-form) weights from rank 0."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """contiguous image range [begin, end) of `rank`; remainders go to the first ranks"""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def broadcast_buffers(buffers, src=0, chunk_bytes=1 << 30):
    """broadcast a list of flat tensors from `src` in <= chunk_bytes pieces (ring broadcast over point-to-point xGMI links is
    per-link bound, so a few large messages are what we want; 1 GiB keeps RCCL's staging modest).  Returns bytes sent."""
    total = 0
    via_host = dist.get_backend() == "gloo"          # CPU-backend rehearsal of the same code path: stage device tensors through host memory
    for t in buffers:
        flat = t.view(-1)
        step = max(1, chunk_bytes // flat.element_size())
        for o in range(0, flat.numel(), step):
            piece = flat[o:o + step]
            if via_host and piece.is_cuda:
                h = piece.cpu()
                dist.broadcast(h, src=src)
                if dist.get_rank() != src:
                    piece.copy_(h)
            else:
                dist.broadcast(piece, src=src)
            total += piece.numel() * piece.element_size()
    return total


def _coll_device(device):
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(n, device):
    """total number of units processed by all ranks"""
    t = torch.tensor([n], dtype=torch.int64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
