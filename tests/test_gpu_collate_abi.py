"""GPU: the multi-GPU helpers of the C ABI -- mi_mcmc_shard_bounds (CPU), mi_mcmc_merge_shards for any world size, and
mi_mcmc_allgather_draws over a real RCCL communicator (world_size 1: the box has one GPU; the N > 1 path of the product is
mcmc_amd.dist under torch.distributed, tests/test_gpu_dist_engine.py)."""
import ctypes as C

import numpy as np
import pytest

import mcmc_amd
from mcmc_amd.dist import shard_bounds


def test_c_shard_bounds_equal_the_python_ones():
    lib = mcmc_amd.lib()
    for total, world in [(65536, 8), (13, 2), (5, 8), (1 << 20, 8), (7, 3), (0, 4)]:
        for r in range(world):
            c0, nl = C.c_uint64(0), C.c_uint64(0)
            lib.mi_mcmc_shard_bounds(C.c_uint64(total), C.c_uint32(world), C.c_uint32(r), C.byref(c0), C.byref(nl))
            assert (c0.value, nl.value) == shard_bounds(total, world, r)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_keep,d,Ct", [(2, 5, 6, 13), (8, 3, 4, 1001), (3, 2, 128, 7), (5, 4, 3, 3), (1, 2, 2, 9)])
def test_merge_shards_interleaves_ragged_rank_major_slabs(world, n_keep, d, Ct):
    import torch
    rng = np.random.default_rng(world)
    full = rng.standard_normal((n_keep, d, Ct))
    parts = []
    for r in range(world):
        c0, nl = shard_bounds(Ct, world, r)
        parts.append(np.ascontiguousarray(full[:, :, c0:c0 + nl]).ravel())
    src = torch.from_numpy(np.concatenate(parts)).cuda()
    dst = torch.zeros((n_keep, d, Ct), dtype=torch.float64, device="cuda")
    rc = mcmc_amd.lib().mi_mcmc_merge_shards(C.c_void_p(src.data_ptr()), C.c_uint32(world), C.c_uint64(n_keep), C.c_uint64(d), C.c_uint64(Ct),
                                            C.c_void_p(dst.data_ptr()), C.c_void_p(0))
    assert rc == 0, mcmc_amd.lib().mi_mcmc_last_error().decode()
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), full)


@pytest.mark.gpu
def test_allgather_draws_over_an_rccl_communicator_of_one_rank():
    import torch
    rccl = C.CDLL("librccl.so.1")
    comm = C.c_void_p(0)
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    n_keep, d, Ct = 4, 8, 50
    local = torch.randn((n_keep, d, Ct), dtype=torch.float64, device="cuda")
    scratch = torch.zeros(n_keep * d * Ct, dtype=torch.float64, device="cuda")
    out = torch.zeros((n_keep, d, Ct), dtype=torch.float64, device="cuda")
    rc = mcmc_amd.lib().mi_mcmc_allgather_draws(comm, C.c_uint32(1), C.c_uint32(0), C.c_void_p(local.data_ptr()), C.c_uint64(n_keep), C.c_uint64(d),
                                               C.c_uint64(Ct), C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(0))
    assert rc == 0, mcmc_amd.lib().mi_mcmc_last_error().decode()
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    rccl.ncclCommDestroy(comm)
