"""Multi-GPU sharding of the MSM (one process per GPU, torch.distributed; backend "nccl" = RCCL).

sum_i s_i P_i is a sum over independent pairs: rank g of G owns the contiguous slice
[g n/G, (g+1) n/G), reduces it to ONE partial group element on its GPU, and the only exchange is an
all-gather of G fixed-size partial records (RCCL has no elliptic-curve reduction operator, so the
"reduce" is gather + local fold).  The compute callbacks are injected so that the exchange logic can
be exercised on CPU with the gloo backend (tests/test_sharded_gloo.py) where the oracle plays the
local compute."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced slice of [0, n) for `rank` of `world` (first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_partials(partial):
    """partial: 1-D tensor (one fixed-size record per rank) -> (world, record) tensor on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return partial.reshape(1, -1)
    world = dist.get_world_size()
    flat = partial.contiguous().reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=partial.dtype, device=partial.device)
    dist.all_gather_into_tensor(out, flat)
    return out.reshape(world, flat.numel())


def sharded_msm(local_partial_fn, fold_fn):
    """local_partial_fn() -> this rank's partial record (tensor); fold_fn(records) -> final result."""
    return fold_fn(all_gather_partials(local_partial_fn()))
