"""Chain sharding across ranks (one process per GPU) and collation of draws_out.

Chains are independent Markov chains (the reference runs one per call, /root/reference/src/hmc.cpp:155-205),
so the path partitions with no data-path collective: rank r owns the global chains
[chain0, chain0 + C_r).  The Philox counter uses the GLOBAL chain id, hence the union of the shards is
bit-identical to one big call.  The only exchange is the optional collation of draws_out: one
all-gather of the [n_keep][d][C_r] slabs (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
import numpy as np


def shard_bounds(n_chains_total, world_size, rank):
    """Contiguous, balanced shards: the first (n % world) ranks get one extra chain."""
    base, extra = divmod(int(n_chains_total), int(world_size))
    c_local = base + (1 if rank < extra else 0)
    chain0 = rank * base + min(rank, extra)
    return chain0, c_local


def run_sharded(algo, kind, init_fn, n_chains_total, settings, runner=None, collate=True, group=None, **target_kw):
    """Every rank samples its shard; draws are all-gathered when `collate`.

    init_fn(chain0, c_local) -> [c_local, d] initial values of the shard's chains.
    runner(algo, kind, init, settings, chain0=..., **target_kw) -> (draws [n_keep, d, C_r], info);
    default: the GPU engine (mcmc_amd.sample).  Returns (draws, n_accept) over ALL chains when
    collating (on every rank), else the local shard.
    """
    import torch
    import torch.distributed as dist

    if runner is None:
        import mcmc_amd
        runner = mcmc_amd.sample
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    chain0, c_local = shard_bounds(n_chains_total, world, rank)
    init = init_fn(chain0, c_local)
    draws, info = runner(algo, kind, init, settings, chain0=chain0, **target_kw)
    n_accept = np.asarray(info["n_accept"], dtype=np.int64)
    if not collate or world == 1:
        return draws, n_accept

    n_keep, d, _ = draws.shape
    c_max = shard_bounds(n_chains_total, world, 0)[1]
    use_cuda = torch.cuda.is_available() and dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    # equal-sized send buffers (all_gather needs them): pad the chain axis of short shards
    send = torch.zeros((n_keep, d, c_max), dtype=torch.float64, device=dev)
    send[:, :, :c_local] = torch.from_numpy(np.ascontiguousarray(draws)).to(dev)
    recv = torch.empty((world * n_keep, d, c_max), dtype=torch.float64, device=dev)   # concatenation along dim 0
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, n_keep, d, c_max)
    acc_send = torch.zeros(c_max, dtype=torch.int64, device=dev)
    acc_send[:c_local] = torch.from_numpy(n_accept).to(dev)
    acc_recv = torch.empty(world * c_max, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(acc_recv, acc_send, group=group)
    acc_recv = acc_recv.view(world, c_max)
    recv, acc_recv = recv.cpu().numpy(), acc_recv.cpu().numpy()
    parts, accs = [], []
    for r in range(world):
        _, c_r = shard_bounds(n_chains_total, world, r)
        parts.append(recv[r][:, :, :c_r])
        accs.append(acc_recv[r][:c_r])
    return np.concatenate(parts, axis=2), np.concatenate(accs)
