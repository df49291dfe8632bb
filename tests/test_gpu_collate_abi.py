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
@pytest.mark.parametrize("fn", ["mi_mcmc_allgather_draws", "mi_mcmc_allgather_draws_ragged"])
def test_allgather_draws_over_an_rccl_communicator_of_one_rank(fn):
    """equal shards: one ncclAllGather; ragged: grouped broadcasts -- both forms, on the communicator a 1-GPU box can build"""
    import torch
    rccl = C.CDLL("librccl.so.1")
    comm = C.c_void_p(0)
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    n_keep, d, Ct = 4, 8, 50
    local = torch.randn((n_keep, d, Ct), dtype=torch.float64, device="cuda")
    scratch = torch.zeros(n_keep * d * Ct, dtype=torch.float64, device="cuda")
    out = torch.zeros((n_keep, d, Ct), dtype=torch.float64, device="cuda")
    rc = getattr(mcmc_amd.lib(), fn)(comm, C.c_uint32(1), C.c_uint32(0), C.c_void_p(local.data_ptr()), C.c_uint64(n_keep), C.c_uint64(d),
                                               C.c_uint64(Ct), C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(0))
    assert rc == 0, mcmc_amd.lib().mi_mcmc_last_error().decode()
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    rccl.ncclCommDestroy(comm)


@pytest.mark.gpu
def test_two_rccl_ranks_on_one_device_is_what_this_box_cannot_do():
    """The N > 1 form of mi_mcmc_allgather_draws needs two GPUs: RCCL refuses two ranks of one communicator on the same device.  This
    test records that fact on the box it runs on (so the claim 'never executed with two RCCL ranks' has a reason attached), and runs
    the collective for real if the box does have two devices."""
    import subprocess, sys, textwrap, os
    import torch
    code = textwrap.dedent("""
        import os, sys, ctypes as C, torch, torch.distributed as dist
        sys.path.insert(0, os.environ["MI_ROOT"])
        import mcmc_amd
        rank, world = int(os.environ["RANK"]), 2
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(rank % ndev)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", rank % ndev))
            x = torch.full((4,), float(rank), device="cuda", dtype=torch.float64)
            out = torch.empty(8, device="cuda", dtype=torch.float64)
            dist.all_gather_into_tensor(out, x)
            torch.cuda.synchronize()
            print("RCCL2 ok", out.tolist())
        except Exception as e:
            print("RCCL2 refused:", type(e).__name__, str(e).replace("\n", " ")[:300])
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_ROOT=root, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", "-c", code]
    # torchrun has no -c: write the script next to the test output instead
    path = "/tmp/mi_rccl2.py"
    open(path, "w").write(code)
    cmd[-2:] = [path]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RCCL2")]
    print("\n".join(lines))
    if not lines and torch.cuda.device_count() < 2:
        # the ranks died inside RCCL / the launcher before Python could catch anything: still "this box cannot", with the evidence
        err = out.stderr.splitlines()
        tail = [l.strip() for l in err if any(k in l for k in ("NCCL", "RCCL", "Duplicate", "duplicate", "invalid usage"))][:6] or err[-6:]
        print("two RCCL ranks on one device: the ranks exited with", out.returncode, "|", " | ".join(tail))
        return
    assert lines, out.stderr[-1500:]
    if torch.cuda.device_count() >= 2:
        assert all(l.startswith("RCCL2 ok") for l in lines)
    else:
        assert any("refused" in l for l in lines), "one device, two ranks: RCCL is expected to refuse (duplicate GPU)"


def test_rank_major_index_addresses_every_chain_of_equal_and_ragged_splits():
    """mi_mcmc_rank_major_index against the layout it documents: shard r packed at n_keep * d * chain0(r) with row length n_local(r)
    (equal shards: [G][n_keep][d][C / G])."""
    lib = mcmc_amd.lib()
    lib.mi_mcmc_rank_major_index.restype = C.c_uint64
    for Ct, world, n_keep, d in [(16, 4, 3, 2), (13, 4, 2, 3), (5, 8, 2, 2), (9, 1, 2, 2)]:
        full = np.arange(n_keep * d * Ct, dtype=np.float64).reshape(n_keep, d, Ct)
        packed = np.concatenate([np.ascontiguousarray(full[:, :, c0:c0 + nl]).ravel() for c0, nl in (shard_bounds(Ct, world, r) for r in range(world))])
        for i in range(n_keep):
            for j in range(d):
                for c in range(Ct):
                    k = lib.mi_mcmc_rank_major_index(C.c_uint64(Ct), C.c_uint32(world), C.c_uint64(n_keep), C.c_uint64(d), C.c_uint64(i), C.c_uint64(j), C.c_uint64(c))
                    assert packed[k] == full[i, j, c]
        assert lib.mi_mcmc_rank_major_index(C.c_uint64(Ct), C.c_uint32(world), C.c_uint64(n_keep), C.c_uint64(d), C.c_uint64(0), C.c_uint64(0), C.c_uint64(Ct)) == 2 ** 64 - 1


@pytest.mark.gpu
def test_rank_major_allgather_and_the_overlapped_begin_wait_form_on_one_rank():
    """The C route of SURVEY 8(e)'s "per kept-draw slab, overlapped with the next trajectory": an hmc run cut into three chunks through
    mi_chains.draw0, every chunk's slab handed to mi_mcmc_allgather_draws_begin while the next chunk samples on the producer stream,
    gathered into ONE rank-major buffer -- equal to the single blocking call over the whole run, bit for bit.  (World size 1: what the
    box can build; with one rank the rank-major layout IS [n_keep][d][C].)"""
    import torch
    lib = mcmc_amd.lib()
    rccl = C.CDLL("librccl.so.1")
    comm = C.c_void_p(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, (C.c_int * 1)(0)) == 0
    d, Ct, n_burn, n_keep = 24, 96, 3, 9
    prec = torch.from_numpy(synth_prec(d)).cuda()
    init = torch.from_numpy(np.ascontiguousarray(np.random.default_rng(1).standard_normal((d, Ct)))).cuda()
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
    st_all = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=n_burn, n_keep_draws=n_keep, n_leap_steps=4, step_size=0.1)
    stream = torch.cuda.current_stream().cuda_stream
    # one blocking run + the rank-major all-gather
    theta = init.clone()
    draws = torch.zeros((n_keep, d, Ct), dtype=torch.float64, device="cuda")
    mcmc_amd.run("hmc", t, st_all, mcmc_amd.make_chains(theta, Ct, draws=draws, mem=mcmc_amd.MEM_DEVICE), stream=stream)
    ref = torch.zeros_like(draws)
    rc = lib.mi_mcmc_allgather_draws_rank_major(comm, C.c_uint32(1), C.c_uint32(0), C.c_void_p(draws.data_ptr()), C.c_uint64(n_keep), C.c_uint64(d),
                                                C.c_uint64(Ct), C.c_void_p(ref.data_ptr()), C.c_void_p(stream))
    assert rc == 0, lib.mi_mcmc_last_error().decode()
    torch.cuda.synchronize()
    assert torch.equal(ref, draws)
    # three chunks, each gathered while the next one samples
    theta = init.clone()
    out = torch.zeros_like(draws)
    bounds, handles, slabs = [0, 3, 7, 9], [], []
    for k in range(3):
        nk = bounds[k + 1] - bounds[k]
        s_k = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=n_burn if k == 0 else 0, n_keep_draws=nk, n_leap_steps=4, step_size=0.1)
        slab = torch.zeros((nk, d, Ct), dtype=torch.float64, device="cuda")
        slabs.append(slab)
        mcmc_amd.run("hmc", t, s_k, mcmc_amd.make_chains(theta, Ct, draws=slab, mem=mcmc_amd.MEM_DEVICE, draw0=0 if k == 0 else n_burn + bounds[k]), stream=stream)
        h = C.c_void_p(0)
        rc = lib.mi_mcmc_allgather_draws_begin(comm, C.c_uint32(1), C.c_uint32(0), C.c_void_p(slab.data_ptr()), C.c_uint64(nk), C.c_uint64(d), C.c_uint64(Ct),
                                               C.c_uint64(bounds[k]), C.c_uint64(n_keep), C.c_void_p(out.data_ptr()), C.c_void_p(stream), C.byref(h))
        assert rc == 0 and h.value, lib.mi_mcmc_last_error().decode()
        handles.append(h)
    for k, h in enumerate(handles):
        assert lib.mi_mcmc_allgather_draws_wait(h, C.c_void_p(stream), C.c_int(k == 2)) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, draws)
    # a row range outside the run is refused, not gathered somewhere
    h = C.c_void_p(0)
    assert lib.mi_mcmc_allgather_draws_begin(comm, C.c_uint32(1), C.c_uint32(0), C.c_void_p(draws.data_ptr()), C.c_uint64(4), C.c_uint64(d), C.c_uint64(Ct),
                                             C.c_uint64(7), C.c_uint64(n_keep), C.c_void_p(out.data_ptr()), C.c_void_p(stream), C.byref(h)) == mcmc_amd.MI_ERR_BAD_ARG
    rccl.ncclCommDestroy(comm)


def synth_prec(d):
    from mcmc_amd import synth
    return synth.dense_gaussian_precision(d, seed=4)
