"""Multi-GPU MSM: point-range sharding and the exchange step.

Each rank reduces its shard of the bases to ONE affine point; the partial points (64..192 bytes each) are
all-gathered as raw bytes (RCCL over xGMI on GPUs, gloo on CPU in the tests) and added on the host.  Elliptic-curve
addition is not a collective reduction operator, so the "reduce" is an all-gather plus local adds -- the same shape
as libff's multi_exp summing its OpenMP chunks.  torch is plumbing here, nothing else."""
from __future__ import annotations

from typing import Tuple

from . import lib as _lib


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """[first, first + count) of the points rank `rank` owns; the first n % world ranks get one extra point."""
    base, extra = divmod(n, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def all_gather_sum(curve: int, group: int, partial, device: str = "cuda"):
    """Sum of every rank's partial point (None = identity); every rank gets the result."""
    import torch
    import torch.distributed as dist

    nbytes = _lib.affine_bytes(curve, group)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return partial
    world = dist.get_world_size()
    mine = torch.frombuffer(bytearray(_lib.encode_point(curve, group, partial)), dtype=torch.uint8).to(device)
    out = torch.empty(nbytes * world, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, mine)
    raw = out.cpu().numpy().tobytes()
    acc = None
    for g in range(world):
        acc = _lib.host_group_op(curve, group, "add", acc, _lib.decode_point(curve, group, raw[nbytes * g : nbytes * (g + 1)]))
    return acc


def all_gather_sum_bytes(curve: int, group: int, partial: bytes, device: str = "cuda") -> bytes:
    """The same exchange on raw affine bytes (plain form, all-zero = identity) with one host call for the sum: the
    per-step path of bench.py, where the exchange must not cost more than a few tens of microseconds of host time."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return partial
    world = dist.get_world_size()
    nbytes = _lib.affine_bytes(curve, group)
    mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(device)
    out = torch.empty(nbytes * world, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, mine)
    return _lib.host_group_sum(curve, group, out.cpu().numpy().tobytes(), world)
