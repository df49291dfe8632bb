"""Chain sharding across ranks (one process per GPU) and collation of draws_out.

Chains are independent Markov chains (the reference runs one per call, /root/reference/src/hmc.cpp:155-205),
so the path partitions with no data-path collective: rank r owns the global chains
[chain0, chain0 + C_r).  The Philox counter uses the GLOBAL chain id, hence the union of the shards is
bit-identical to one big call.  The only exchange is the optional collation of draws_out: one
all-gather of the [n_keep][d][C_r] slabs -- RCCL over xGMI on GPUs, fed from and received into HBM
(no host round trip); gloo in the CPU tests.
"""
import ctypes
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


def _copy_settings(settings, **fields):
    """A by-value copy of a ctypes mi_settings (the arrays it points to stay owned by the original)."""
    c = type(settings).from_buffer_copy(settings)
    c._keep = getattr(settings, "_keep", None)
    for k, v in fields.items():
        setattr(c, k, v)
    return c


def _gather_start(local, c_local, n_chains_total, world, group, comm_dev):
    """The asynchronous half of _gather_ragged: returns (work, recv, send) -- the pending collective, its receive buffer [world, ..., c_max]
    and the send buffer it reads (kept alive until the wait); finish with _gather_finish."""
    import torch
    import torch.distributed as dist
    c_max = shard_bounds(n_chains_total, world, 0)[1]
    lead = tuple(local.shape[:-1])
    send = torch.zeros(lead + (c_max,), dtype=local.dtype, device=comm_dev)
    if c_local:
        send[..., :c_local] = local.to(comm_dev)
    recv = torch.empty((world,) + lead + (c_max,), dtype=local.dtype, device=comm_dev)
    flat = recv.view((world * lead[0],) + lead[1:] + (c_max,)) if lead else recv.view(world * c_max)
    work = dist.all_gather_into_tensor(flat, send, group=group, async_op=True)
    return work, recv, send


def _gather_finish(started, n_chains_total, world):
    import torch
    work, recv, _send = started
    work.wait()
    parts = [recv[r][..., :shard_bounds(n_chains_total, world, r)[1]] for r in range(world)]
    return torch.cat(parts, dim=-1)


def run_sharded_overlapped(algo, kind, init_fn, n_chains_total, settings, n_chunks, runner=None, group=None, device=None, **target_kw):
    """run_sharded with the collation OVERLAPPED with the sampling ("or per kept-draw slab, overlapped with the next
    trajectory"): the kept draws are produced in n_chunks consecutive calls chained through mi_chains.draw0 (bit-identical to one
    call: tests/test_gpu_resume.py), and the all-gather of chunk k's slab is issued asynchronously -- RCCL runs it on its own stream --
    while chunk k + 1 samples.  hmc / mala / rwmh, and nuts when its adaptation window lies inside the burn-in (a nuts continuation must
    start after it and takes the adapted step sizes back in).  runner: as in run_sharded, and it must accept draw0= (and step_size_in=
    for nuts) and return info['theta'] = the chains' last state [d, C_r].  Returns (draws [n_keep, d, C], n_accept [C]) on every rank."""
    import torch
    import torch.distributed as dist

    n_keep, n_burn = int(settings.n_keep_draws), int(settings.n_burnin_draws)
    n_chunks = max(1, min(int(n_chunks), n_keep))
    if algo == "nuts" and int(settings.n_adapt_draws) > n_burn:
        raise ValueError("run_sharded_overlapped: nuts needs n_adapt_draws <= n_burnin_draws (a continuation starts after the adaptation window)")
    if algo == "rmhmc":
        raise ValueError("run_sharded_overlapped: rmhmc is not chained through draw0")
    engine = runner is None
    dev = bind_device(device) if engine else None
    if engine:
        import mcmc_amd
        if dev is None:
            raise mcmc_amd.MiMcmcError(mcmc_amd.MI_ERR_NO_DEVICE, "run_sharded_overlapped: no GPU visible and no runner given (no CPU path)")
        runner = lambda a, k, init, st, chain0=0, **kw: mcmc_amd.sample_device(a, k, init, st, chain0=chain0, device=dev, **kw)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    chain0, c_local = shard_bounds(n_chains_total, world, rank)
    nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
    comm_dev = dev if (nccl and dev is not None) else torch.device("cpu")
    bounds = [(n_keep * i) // n_chunks for i in range(n_chunks + 1)]
    state = init_fn(chain0, c_local) if c_local > 0 else None
    eps_in = None
    acc = None
    started, locals_ = [], []
    d = None
    for i in range(n_chunks):
        s_i = _copy_settings(settings, n_burnin_draws=(n_burn if i == 0 else 0), n_keep_draws=bounds[i + 1] - bounds[i])
        kw = dict(target_kw)
        if i > 0:
            kw["draw0"] = n_burn + bounds[i]
            if algo == "nuts":
                kw["step_size_in"] = eps_in
        slab = None
        if c_local > 0:
            slab, info = runner(algo, kind, state, s_i, chain0=chain0, **kw)
            th = info["theta"]
            state = th.t() if hasattr(th, "t") and not isinstance(th, np.ndarray) else np.ascontiguousarray(np.asarray(th).T)
            eps_in = info.get("eps")
            a = info["n_accept"]
            a = a if hasattr(a, "device") and not isinstance(a, np.ndarray) else torch.from_numpy(np.asarray(a).astype(np.int64))
            acc = a.clone() if acc is None else acc + a
            slab = slab if hasattr(slab, "device") and not isinstance(slab, np.ndarray) else torch.from_numpy(np.ascontiguousarray(slab))
            d = slab.shape[1]
        if world == 1:
            locals_.append(slab)
            continue
        if d is None or i == 0:                       # the slab's chain-independent shape: from rank 0, which always has chains
            d_t = torch.tensor([d or 0], dtype=torch.int64).to(comm_dev)
            dist.broadcast(d_t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            d = int(d_t.item())
        if slab is None:
            slab = torch.zeros((bounds[i + 1] - bounds[i], d, 0), dtype=torch.float64, device=comm_dev)
        started.append(_gather_start(slab, c_local, n_chains_total, world, group, comm_dev))    # runs while the next chunk samples
    if world == 1:
        draws = torch.cat(locals_, dim=0)
        return (draws, acc) if engine else (draws.numpy(), acc.numpy())
    if acc is None:
        acc = torch.zeros(0, dtype=torch.int64, device=comm_dev)
    all_draws = torch.cat([_gather_finish(st_, n_chains_total, world) for st_ in started], dim=0)
    all_acc = _gather_ragged(acc, c_local, n_chains_total, world, group, comm_dev)
    if engine:
        return all_draws.to(dev), all_acc.to(dev)
    return all_draws.numpy(), all_acc.numpy()


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


# ---- the C-ABI collation (include/mi_mcmc.h: mi_mcmc_allgather_draws_rank_major, _begin / _wait) under a torch.distributed launch ----------------
# The C entry points take the caller's ncclComm_t.  A torch process group does not hand its communicator out, so a rank builds one next to it:
# rank 0 makes the ncclUniqueId, the process group (any backend) carries its 128 bytes to the others, every rank calls ncclCommInitRank --
# in the librccl torch has already mapped (the engine's dlopen("librccl.so.1") resolves to the same image).

class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


class RcclComm:
    """An RCCL communicator of this rank's process group for the C ABI's collation calls.  `solo=True`: a ONE-rank communicator on the
    current device (what a box with one GPU can build: two ranks of one communicator on one device are refused by RCCL)."""

    def __init__(self, group=None, solo=False):
        import ctypes as C
        import torch
        import torch.distributed as dist
        self._rccl = C.CDLL("librccl.so.1")
        self.comm = C.c_void_p(0)
        if solo or not dist.is_initialized() or dist.get_world_size(group) == 1:
            self.world, self.rank = 1, 0
            dev = (C.c_int * 1)(torch.cuda.current_device())
            rc = self._rccl.ncclCommInitAll(C.byref(self.comm), 1, dev)
        else:
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
            uid = _NcclUniqueId()
            if self.rank == 0:
                rc0 = self._rccl.ncclGetUniqueId(C.byref(uid))
                if rc0 != 0:
                    raise RuntimeError(f"ncclGetUniqueId failed ({rc0})")
            box = [bytes(uid.internal) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            C.memmove(C.byref(uid), box[0], 128)
            self._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
            rc = self._rccl.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
        if rc != 0 or not self.comm.value:
            raise RuntimeError(f"RCCL communicator of {self.world} rank(s) could not be built (ncclResult {rc})")

    def close(self):
        if self.comm is not None and self.comm.value:
            self._rccl.ncclCommDestroy(self.comm)
        self.comm = None


def collate_rank_major(comm, local_draws, n_chains_total, out=None, stream=None):
    """mi_mcmc_allgather_draws_rank_major: this rank's slab [n_keep][d][n_local] (HBM) -> all_rank_major (HBM; SURVEY 8(e)'s receive layout
    [G][n_keep][d][C / G] for equal shards, packed shard after shard otherwise).  Enqueued on `stream`; returns the flat receive tensor."""
    import ctypes as C
    import torch
    import mcmc_amd
    n_keep, d = int(local_draws.shape[0]), int(local_draws.shape[1])
    if out is None:
        out = torch.empty(n_keep * d * int(n_chains_total), dtype=torch.float64, device=local_draws.device)
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    mcmc_amd._check(mcmc_amd.lib().mi_mcmc_allgather_draws_rank_major(
        comm.comm, C.c_uint32(comm.world), C.c_uint32(comm.rank), C.c_void_p(local_draws.data_ptr()), C.c_uint64(n_keep), C.c_uint64(d),
        C.c_uint64(int(n_chains_total)), C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    return out


def collate_begin(comm, slab, n_chains_total, row0, n_keep_total, out, producer_stream=None):
    """mi_mcmc_allgather_draws_begin: the gather of kept rows [row0, row0 + slab.shape[0]) is ordered behind the producer stream's work so far
    and runs on the library's communication stream; returns the handle for collate_wait.  The caller keeps `slab` alive and untouched."""
    import ctypes as C
    import torch
    import mcmc_amd
    producer_stream = torch.cuda.current_stream().cuda_stream if producer_stream is None else producer_stream
    h = C.c_void_p(0)
    mcmc_amd._check(mcmc_amd.lib().mi_mcmc_allgather_draws_begin(
        comm.comm, C.c_uint32(comm.world), C.c_uint32(comm.rank), C.c_void_p(slab.data_ptr()), C.c_uint64(int(slab.shape[0])),
        C.c_uint64(int(slab.shape[1])), C.c_uint64(int(n_chains_total)), C.c_uint64(int(row0)), C.c_uint64(int(n_keep_total)),
        C.c_void_p(out.data_ptr()), C.c_void_p(producer_stream), C.byref(h)))
    return h


def collate_wait(handle, consumer_stream=None, block_host=False):
    import ctypes as C
    import torch
    import mcmc_amd
    consumer_stream = torch.cuda.current_stream().cuda_stream if consumer_stream is None else consumer_stream
    mcmc_amd._check(mcmc_amd.lib().mi_mcmc_allgather_draws_wait(handle, C.c_void_p(consumer_stream), C.c_int(1 if block_host else 0)))
