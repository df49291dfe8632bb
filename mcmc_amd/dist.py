"""Chain sharding across ranks (one process per GPU) and collation of draws_out.

Chains are independent Markov chains (the reference runs one per call, /root/reference/src/hmc.cpp:155-205),
so the path partitions with no data-path collective: rank r owns the global chains
[chain0, chain0 + C_r).  The Philox counter uses the GLOBAL chain id, hence the union of the shards is
bit-identical to one big call.  The only exchange is the optional collation of draws_out: one
all-gather of the [n_keep][d][C_r] slabs -- RCCL over xGMI on GPUs, fed from and received into HBM
(no host round trip); gloo in the CPU tests.
"""
import os

import numpy as np


def shard_bounds(n_chains_total, world_size, rank):
    """Contiguous, balanced shards: the first (n % world) ranks get one extra chain."""
    base, extra = divmod(int(n_chains_total), int(world_size))
    c_local = base + (1 if rank < extra else 0)
    chain0 = rank * base + min(rank, extra)
    return chain0, c_local


def bind_device(device=None):
    """One process per GPU: select this rank's device (LOCAL_RANK under torchrun) before any engine call.  Returns the
    torch.device, or None when no GPU is visible (CPU tests with an explicit runner)."""
    import torch
    if not torch.cuda.is_available():
        return None
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    torch.cuda.set_device(dev)
    return dev


def _gather_ragged(local, c_local, n_chains_total, world, group, comm_dev):
    """All-gather tensors whose LAST axis is the (ragged) chain axis; returns the concatenation over ranks."""
    import torch
    import torch.distributed as dist
    c_max = shard_bounds(n_chains_total, world, 0)[1]
    lead = tuple(local.shape[:-1])
    send = torch.zeros(lead + (c_max,), dtype=local.dtype, device=comm_dev)     # equal-sized buffers: pad short shards
    if c_local:
        send[..., :c_local] = local.to(comm_dev)
    recv = torch.empty((world,) + lead + (c_max,), dtype=local.dtype, device=comm_dev)
    flat = recv.view((world * lead[0],) + lead[1:] + (c_max,)) if lead else recv.view(world * c_max)   # concatenation along dim 0
    dist.all_gather_into_tensor(flat, send, group=group)
    parts = [recv[r][..., :shard_bounds(n_chains_total, world, r)[1]] for r in range(world)]
    return torch.cat(parts, dim=-1)


def run_sharded(algo, kind, init_fn, n_chains_total, settings, runner=None, collate=True, group=None, device=None,
                **target_kw):
    """Every rank samples its shard; draws are all-gathered when `collate`.

    init_fn(chain0, c_local) -> [c_local, d] initial values of the shard's chains.
    runner: None = the GPU engine, device-resident (mcmc_amd.sample_device on this rank's GPU; results are torch tensors
    in HBM and the all-gather runs on them directly); or a callable
    runner(algo, kind, init, settings, chain0=..., **target_kw) -> (draws [n_keep, d, C_r], info) returning numpy arrays
    (the CPU tests pass the oracle).  A rank whose shard is empty (world_size > n_chains_total) runs nothing but still
    joins the collectives.  Returns (draws [n_keep, d, C], n_accept [C]) over ALL chains when collating (on every rank),
    else the local shard.
    """
    import torch
    import torch.distributed as dist

    engine = runner is None
    dev = bind_device(device) if engine else None
    if engine:
        import mcmc_amd
        if dev is None:
            raise mcmc_amd.MiMcmcError(mcmc_amd.MI_ERR_NO_DEVICE, "run_sharded: no GPU visible and no runner given (no CPU path)")
        runner = lambda a, k, init, st, chain0=0, **kw: mcmc_amd.sample_device(a, k, init, st, chain0=chain0, device=dev, **kw)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    chain0, c_local = shard_bounds(n_chains_total, world, rank)
    n_keep = int(settings.n_keep_draws)
    draws = n_accept = None
    if c_local > 0:
        init = init_fn(chain0, c_local)
        draws, info = runner(algo, kind, init, settings, chain0=chain0, **target_kw)
        n_accept = info["n_accept"]
    if not engine:                      # numpy in, numpy out
        draws = None if draws is None else torch.from_numpy(np.ascontiguousarray(draws))
        n_accept = None if n_accept is None else torch.from_numpy(np.asarray(n_accept).astype(np.int64))
    if not collate or world == 1:
        if engine:
            return draws, n_accept
        # an empty shard (world_size > n_chains_total) hands back nothing rather than tripping over None
        return (None if draws is None else draws.numpy()), (None if n_accept is None else n_accept.numpy())

    nccl = dist.get_backend(group) == "nccl"
    comm_dev = dev if (nccl and dev is not None) else torch.device("cpu")
    # the chain-independent shape of a slab: from the local result, or (empty shard) from the first rank, which always has chains
    d_t = torch.tensor([draws.shape[1] if draws is not None else 0], dtype=torch.int64).to(comm_dev)
    dist.broadcast(d_t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    d = int(d_t.item())
    if draws is None:
        draws = torch.zeros((n_keep, d, 0), dtype=torch.float64, device=comm_dev)
        n_accept = torch.zeros(0, dtype=torch.int64, device=comm_dev)
    if n_keep == 0:                     # nothing was kept: there is no slab to gather, only the accept counts
        all_acc = _gather_ragged(n_accept, c_local, n_chains_total, world, group, comm_dev)
        empty = torch.zeros((0, d, n_chains_total), dtype=torch.float64, device=comm_dev)
        return (empty.to(dev), all_acc.to(dev)) if engine else (empty.numpy(), all_acc.numpy())
    all_draws = _gather_ragged(draws, c_local, n_chains_total, world, group, comm_dev)
    all_acc = _gather_ragged(n_accept, c_local, n_chains_total, world, group, comm_dev)
    if engine:
        return all_draws.to(dev), all_acc.to(dev)
    return all_draws.numpy(), all_acc.numpy()
