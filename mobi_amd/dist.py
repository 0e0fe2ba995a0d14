"""Data-parallel sharding of inpainting objects over the GPUs of one node.

Objects are independent; the camera/lidar pair of an object (and its classifier-free-guidance
twin) always stays on one rank.  Partition = contiguous blocks of objects per rank, weights
replicated, no communication during sampling; ONE collective per batch: the all-gather of the
decoded images (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  The reference has no multi-GPU inference (inference_test_bench.py:337 is one process, one GPU).
"""
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_objects: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of objects for `rank`; the first `n % world` ranks take one more."""
    q, r = divmod(n_objects, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, n_objects: int, rank: int = None, world_size: int = None):
    """Slice every tensor of a (nested) batch dict whose leading dim is the object axis."""
    if rank is None:
        rank, world_size = world()
    lo, hi = shard_range(n_objects, rank, world_size)

    def cut(x):
        if isinstance(x, dict):
            return {k: cut(v) for k, v in x.items()}
        if isinstance(x, torch.Tensor) and x.dim() > 0 and x.shape[0] == n_objects:
            return x[lo:hi]
        return x

    return cut(batch)


def gather_objects(local: torch.Tensor, n_objects: int) -> torch.Tensor:
    """All-gather per-rank results `[b_rank, ...]` into `[n_objects, ...]` on every rank, in object order.
    Ragged shards (n % world != 0) are padded to the largest shard for the collective and trimmed after."""
    rank, ws = world()
    if ws == 1:
        return local
    sizes = [shard_range(n_objects, r, ws)[1] - shard_range(n_objects, r, ws)[0] for r in range(ws)]
    bmax = max(sizes)
    if local.shape[0] < bmax:
        pad = torch.zeros((bmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((ws * bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if all(s == bmax for s in sizes):
        return out
    return torch.cat([out[r * bmax: r * bmax + sizes[r]] for r in range(ws)], dim=0)


def gather_decoded(images: Dict[str, torch.Tensor], n_objects: int) -> Dict[str, torch.Tensor]:
    """The per-batch collective of the path: decoded camera `[B,3,R,R]` and range `[B,2,R,R]` images."""
    return {k: gather_objects(v, n_objects) for k, v in images.items()}


def check_same_layout(names: List[str], numels: List[int], device) -> None:
    """Every rank must bring the SAME tensors in the same order to a bucketed collective: buckets are cut by name order and
    size, so a rank with another key set would all-reduce buffers of another length (a hang on RCCL, an error on gloo) or sum
    different parameters into each other.  One tiny collective up front: the MAX and the MIN over the ranks of a 62-bit digest
    of (name, numel)* must agree; raises on every rank otherwise."""
    import hashlib
    rank, ws = world()
    if ws == 1:
        return
    h = hashlib.sha1(";".join(f"{k}:{n}" for k, n in zip(names, numels)).encode()).digest()
    d = int.from_bytes(h[:8], "big") >> 2
    t = torch.tensor([d, -d], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0]) != -int(t[1]):
        raise RuntimeError(f"rank {rank}: the ranks hold different gradient sets ({len(names)} tensors here) -- every rank must pass "
                           f"the same names and sizes to allreduce_gradients (zeros for a branch it did not take)")


def allreduce_gradients(grads: Dict[str, torch.Tensor], bucket_bytes: int = 256 << 20, average: bool = True) -> Dict[str, torch.Tensor]:
    """The gradient collective of the training step (the reference wraps the model in DDP, main.py:510 -- Lightning's
    `ddp` strategy): every rank holds the gradients of ITS objects; after the call every rank holds their sum (or mean).
    The tensors are flattened in name order into fp32 buckets of about `bucket_bytes` and each bucket is ONE all-reduce --
    xGMI is point-to-point, a ring all-reduce is bound by one link (~153 GB/s), so few large transfers beat one per tensor
    (432 adapter tensors, ~180 M parameters = 720 MB fp32: three buckets by default).  In place; returns `grads`.
    Backend "nccl" is RCCL on ROCm; gloo in the CPU tests."""
    rank, ws = world()
    if ws == 1:
        return grads
    names = sorted(grads)
    check_same_layout(names, [grads[k].numel() for k in names], grads[names[0]].device if names else torch.device("cpu"))
    i = 0
    while i < len(names):
        bucket, size = [], 0
        while i < len(names) and (not bucket or size + grads[names[i]].numel() * 4 <= bucket_bytes):
            bucket.append(names[i])
            size += grads[names[i]].numel() * 4
            i += 1
        flat = torch.cat([grads[k].reshape(-1).float() for k in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= ws
        off = 0
        for k in bucket:
            n = grads[k].numel()
            grads[k].copy_(flat[off:off + n].view_as(grads[k]))
            off += n
    return grads
