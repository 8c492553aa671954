"""Multi-rank path on CPU: world_size 2 (and 4) over gloo (127.0.0.1).  The *host* code under test is the product's
(mnk.dist all-reduces inside BNActFn, GradAverager, shard_batch, TrainStep); the kernels run on the CPU emulator
build so no GPU is needed.  Property: 2 ranks x B/2 samples (SyncBN + gradient averaging) == 1 rank x B samples."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "monkey-net_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _bring_up_exchange(lib, mdist, rank, world):
    """The peer-to-peer SyncBN exchange between the PROCESSES of this test (TEST INFRASTRUCTURE mirroring mnk.dist.p2p_comm, which
    needs a device): the emulator's mailboxes are POSIX shared memory, its IPC handles their names (tests/hipemu/hipemu.cpp) --
    create, export, gather, connect, self-test, then hand the handle to mnk.dist as p2p_comm would."""
    import ctypes
    h = ctypes.c_void_p()
    mine = (ctypes.c_ubyte * 64)()
    lib.call("mnk_p2p_create", rank, world, ctypes.byref(h))
    lib.call("mnk_p2p_export", h, mine)
    gathered = [None] * world
    dist.all_gather_object(gathered, bytes(mine))
    lib.call("mnk_p2p_connect", h, ctypes.c_char_p(b"".join(gathered)))
    dist.barrier()
    probe = torch.arange(1, 65, dtype=torch.float32) * float(rank + 1)
    got = torch.empty_like(probe)
    lib.call("mnk_p2p_allreduce", h, probe.data_ptr(), got.data_ptr(), 64, 20000, 0)
    assert torch.equal(got, torch.arange(1, 65, dtype=torch.float32) * float(world * (world + 1) // 2))
    mdist.P2P_TIMEOUT_MS = 60000
    mdist._P2P.update(tried=True, handle=h.value, max=int(lib.query("mnk_p2p_max_floats")))
    return h


def _worker(rank, world, port, emu_path, out_dir, mnk_adam=False, p2p=False):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from mnk import _lib, dist as mdist, engine
    import _util
    from oracle import cases
    from test_modules import build
    lib = _util.set_library(emu_path, strict=False)
    calls = {"fwd": 0, "bwd": 0, "general": 0}
    if p2p:
        from mnk import ops
        exchange = _bring_up_exchange(lib, mdist, rank, world)
        real = ops._call

        def counting(name, *a):
            key = "fwd" if name == "mnk_bn_small_fwd_sync" else "bwd" if name == "mnk_bn_small_bwd_sync" else \
                "general" if name.endswith("_sync") else None
            if key:
                calls[key] += 1
            return real(name, *a)

        ops._call = counting
    cfg = cases.TINY2
    gen, disc, kpd = build(cfg)
    for i, m in enumerate((gen, disc, kpd)):
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 7 + i)
        m.load_state_dict(sd)
    src, drv = cases.smooth_pair(max(4, world), 32, 32)
    x = {"source": mdist.shard_batch(src).contiguous(), "video": mdist.shard_batch(drv).contiguous()}
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=mnk_adam)
    g_losses, d_losses, _ = step.step(x)
    # the losses are per-shard means: average them over ranks like the reference's gather + mean (train.py:114)
    lv = torch.tensor([float(v) for v in g_losses + d_losses], dtype=torch.float64)
    dist.all_reduce(lv)
    lv /= world
    if p2p:
        mdist.check_p2p()
        assert calls["fwd"] > 0 and calls["bwd"] > 0 and calls["general"] > 0, calls
    torch.save({"losses": lv, "gen": {k: v.clone() for k, v in gen.state_dict().items()},
                "kp": {k: v.clone() for k, v in kpd.state_dict().items()},
                "disc": {k: v.clone() for k, v in disc.state_dict().items()}},
               os.path.join(out_dir, "rank%d.pt" % rank))
    # plain collectives of mnk.dist
    t = torch.full((3,), float(rank + 1))
    mdist.all_reduce_sum_(t)
    tri = world * (world + 1) / 2.0
    assert torch.equal(t, torch.full((3,), tri))
    mdist.check_equal_shards(5)
    try:
        mdist.check_equal_shards(5 + rank)
        raise AssertionError("unequal shards must be refused")
    except ValueError as e:
        assert "different numbers of samples" in str(e)
    # the shard of ONE rank changes (a last partial batch): rank 0 keeps the batch size it was checked with, the
    # others get one sample less.  The check must be entered by every rank -- all of them raise; before, rank 0 skipped the
    # collective (its size was cached) and walked into the SyncBN all-reduces while the others sat in the check: a hang
    xs = {k: (v if rank == 0 else v[:-1]).contiguous() for k, v in x.items()}
    try:
        step.step(xs)
        raise AssertionError("a shard that changed on some ranks only must be refused on every rank")
    except ValueError as e:
        assert "different numbers of samples" in str(e)
    sums, count = mdist.combine_bn_stats(torch.tensor([1.0 * (rank + 1), 2.0]), 10)
    assert count == 10 * world and torch.equal(sums, torch.tensor([tri, 2.0 * world]))
    if p2p:
        dist.barrier()                       # nobody unmaps a mailbox a peer may still be reading
        mdist.disable_p2p()
        lib.call("mnk_p2p_destroy", exchange)
    dist.destroy_process_group()


def _single(emu_path, mnk_adam=False, batch=4):
    _setup_paths()
    from mnk import _lib, engine
    import _util
    from oracle import cases
    from test_modules import build
    _util.set_library(emu_path, strict=False)
    cfg = cases.TINY2
    gen, disc, kpd = build(cfg)
    for i, m in enumerate((gen, disc, kpd)):
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 7 + i)
        m.load_state_dict(sd)
    src, drv = cases.smooth_pair(batch, 32, 32)
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=mnk_adam)
    g_losses, d_losses, _ = step.step({"source": src, "video": drv})
    out = {"losses": torch.tensor([float(v) for v in g_losses + d_losses], dtype=torch.float64),
           "gen": gen.state_dict(), "kp": kpd.state_dict(), "disc": disc.state_dict()}
    _util.set_library(None)
    return out


@pytest.mark.parametrize("mnk_adam,world,p2p", [(False, 2, False), (True, 2, False), (True, 4, False), (True, 2, True), (True, 4, True),
                                                (True, 8, True)],
                         ids=["torch-adam+GradAverager", "mnk-adam-flat-buffer", "mnk-adam-flat-buffer-4-ranks",
                              "peer-to-peer-syncbn-2-ranks", "peer-to-peer-syncbn-4-ranks", "peer-to-peer-syncbn-8-ranks"])
def test_ranks_equal_one_rank_big_batch(mnk_adam, world, p2p):
    """p2p: the SyncBN sums of the ranks travel through the library's own exchange -- the kernels of csrc/p2p.hip and the *_sync
    kernels of batchnorm.hip, mailboxes mapped between the PROCESSES of the test (shared memory standing in for IPC-mapped HBM) --
    instead of torch.distributed all-reduces: the multi-rank protocol (slots, sequence numbers, rank-ordered sums, the exchange
    inside the statistics' second stage and inside the one-launch small-layer kernels) runs in the CPU suite."""
    from conftest import emu_library_path
    from oracle import cases
    emu = emu_library_path()
    ref = _single(emu, mnk_adam, max(4, world))
    with tempfile.TemporaryDirectory() as tmp:
        port = 29500 + (os.getpid() % 2000) + (1 if mnk_adam else 0) + world + (7 if p2p else 0)
        mp.spawn(_worker, args=(world, port, emu, tmp, mnk_adam, p2p), nprocs=world, join=True)
        r0 = torch.load(os.path.join(tmp, "rank0.pt"), weights_only=False)
        r1 = torch.load(os.path.join(tmp, "rank1.pt"), weights_only=False)
    # ranks stay bit-identical replicas after the step (same averaged gradients, same all-reduced BN statistics)
    for key in ("gen", "kp", "disc"):
        for k in r0[key]:
            assert torch.equal(r0[key][k], r1[key][k]), (key, k)
    # and the 2 x B/2 job matches the 1 x B job: losses, running statistics (SyncBN) and updated parameters
    assert float((r0["losses"] - ref["losses"]).abs().max()) < 5e-5 * float(ref["losses"].abs().max() + 1)
    lr = cases.TINY2["train_params"]["lr"]
    for key in ("gen", "kp", "disc"):
        for k, v in ref[key].items():
            if "running" in k:
                assert float((r0[key][k] - v).abs().max()) < 1e-5 * (1 + float(v.abs().max())), (key, k)
            elif v.is_floating_point() and not cases.is_noise_bias(k):
                # updated parameters: one Adam step moves every element by ~lr * sign(gradient); the sum over two shards
                # differs from the one-batch sum by rounding, which can flip the sign of a near-zero gradient element
                d = (r0[key][k] - v).abs()
                assert float(d.max()) <= 2.1 * lr, (key, k, float(d.max()))
                flipped = int((d > 0.05 * lr).sum())          # (one element of a 32-element vector is already 3.1 %)
                assert flipped <= max(2, 0.03 * d.numel()), (key, k, flipped, d.numel())


def _forced_single_rank_worker(rank, port, emu_path, out_dir):
    """one process, a gloo group of world size 1 with MNK_DIST_FORCE=1: every multi-rank branch of the host code runs, and the
    SyncBN sums go through the kernels that carry the peer-to-peer exchange -- on a one-rank handle that this TEST builds (the
    product's own bring-up, mnk.dist.p2p_comm, needs a device and peers to map; the emulator has neither)"""
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", MNK_DIST_FORCE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.set_num_threads(2)
    import ctypes
    from mnk import dist as mdist, engine, ops
    import _util
    from oracle import cases
    from test_modules import build
    lib = _util.set_library(emu_path, strict=False)
    h = ctypes.c_void_p()
    lib.call("mnk_p2p_create", 0, 1, ctypes.byref(h))
    mdist._P2P.update(tried=True, handle=h.value, max=int(lib.query("mnk_p2p_max_floats")))
    assert mdist.active() and ops._sync_handle(8) is not None
    cfg = cases.TINY2
    gen, disc, kpd = build(cfg)
    for i, m in enumerate((gen, disc, kpd)):
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 7 + i)
        m.load_state_dict(sd)
    src, drv = cases.smooth_pair(4, 32, 32)
    calls = {"fwd": 0, "bwd": 0, "general": 0}
    real = ops._call

    def counting(name, *a):
        if name == "mnk_bn_small_fwd_sync":
            calls["fwd"] += 1
        elif name == "mnk_bn_small_bwd_sync":
            calls["bwd"] += 1
        elif name.endswith("_sync"):
            calls["general"] += 1
        return real(name, *a)

    ops._call = counting
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=True)
    g_losses, d_losses, _ = step.step({"source": src, "video": drv})
    mdist.check_p2p()
    assert calls["fwd"] > 0 and calls["bwd"] > 0 and calls["general"] > 0, calls
    assert not ops._REDUCED, "every 'already reduced' mark must have been consumed"
    torch.save({"losses": torch.tensor([float(v) for v in g_losses + d_losses], dtype=torch.float64), "calls": calls,
                "gen": gen.state_dict(), "kp": kpd.state_dict(), "disc": disc.state_dict()}, os.path.join(out_dir, "forced.pt"))
    dist.destroy_process_group()


def test_forced_single_rank_iteration_runs_through_the_exchange_kernels_and_equals_the_plain_one():
    from conftest import emu_library_path
    from oracle import cases
    emu = emu_library_path()
    ref = _single(emu, True)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_forced_single_rank_worker, args=(29500 + (os.getpid() % 2000) + 17, emu, tmp), nprocs=1, join=True)
        got = torch.load(os.path.join(tmp, "forced.pt"), weights_only=False)
    # one rank's exchange adds 0.f + x: the sums are the plain ones; the small layers run another thread map (rounding)
    assert float((got["losses"] - ref["losses"]).abs().max()) < 2e-5 * float(ref["losses"].abs().max() + 1)
    lr = cases.TINY2["train_params"]["lr"]
    for key in ("gen", "kp", "disc"):
        for k, v in ref[key].items():
            if "running" in k:
                assert float((got[key][k] - v).abs().max()) < 1e-5 * (1 + float(v.abs().max())), (key, k)
            elif v.is_floating_point() and not cases.is_noise_bias(k):
                d = (got[key][k] - v).abs()
                assert float(d.max()) <= 2.1 * lr, (key, k, float(d.max()))
                assert int((d > 0.05 * lr).sum()) <= max(2, 0.03 * d.numel()), (key, k)


def _absent_peer_worker(rank, world, port, emu_path, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    from mnk import dist as mdist
    import _util
    lib = _util.set_library(emu_path, strict=False)
    h = _bring_up_exchange(lib, mdist, rank, world)
    mdist.P2P_TIMEOUT_MS = 300
    dist.barrier()
    if rank == 0:                                     # rank 1 stays away from the next two exchanges
        x = torch.arange(1, 9, dtype=torch.float32)
        t0 = time.perf_counter()
        bad = torch.empty_like(x)
        lib.call("mnk_p2p_allreduce", h, x.data_ptr(), bad.data_ptr(), 8, mdist.P2P_TIMEOUT_MS, 0)
        t1 = time.perf_counter()
        again = torch.empty_like(x)
        lib.call("mnk_p2p_allreduce", h, x.data_ptr(), again.data_ptr(), 8, mdist.P2P_TIMEOUT_MS, 0)
        t2 = time.perf_counter()
        assert bool(torch.isnan(bad).all()) and bool(torch.isnan(again).all()), (bad, again)     # never a stale mailbox word
        assert mdist.p2p_error() == 2                                                              # 1 + the absent rank
        assert t1 - t0 > 0.25 and t2 - t1 < 0.2, (t1 - t0, t2 - t1)                                # waited once, not twice
        try:
            mdist.check_p2p()
            raise AssertionError("check_p2p() must raise after a give-up")
        except RuntimeError as e:
            assert "rank 1" in str(e)
        # a user-owned loop never reaches TrainStep's poll -- the checkpoint helpers and the wrapper's forward raise too
        from sync_batchnorm import SynchronizedBatchNorm3d, DataParallelWithCallback
        from mnk.optim import MnkAdam
        mdist._P2P["handle"] = h.value
        bn = SynchronizedBatchNorm3d(4)
        net = torch.nn.Sequential(torch.nn.Conv3d(3, 4, 1), bn)
        for what in (net.state_dict, MnkAdam(net.parameters(), lr=1e-3).state_dict):
            try:
                what()
                raise AssertionError("state_dict() must raise after a give-up")
            except RuntimeError as e:
                assert "rank 1" in str(e)
        wrapper = DataParallelWithCallback(torch.nn.Identity())
        wrapper._calls = 31
        try:
            wrapper(torch.zeros(1))
            raise AssertionError("the wrapper's 32nd call must raise after a give-up")
        except RuntimeError as e:
            assert "rank 1" in str(e)
        mdist._P2P["handle"] = None
        assert net.state_dict() is not None          # (no exchange connected: nothing is polled)
    dist.barrier()
    mdist.disable_p2p()
    lib.call("mnk_p2p_destroy", h)
    dist.destroy_process_group()


def test_a_peer_that_stays_away_poisons_the_sums_and_is_reported():
    """ADVICE r4 (also tests/test_p2p_gpu.py on the device): an exchange that gives a peer up returns NaN, sets the error word,
    does not wait the timeout a second time, and mnk.dist.check_p2p() raises"""
    from conftest import emu_library_path
    emu = emu_library_path()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_absent_peer_worker, args=(2, 29500 + (os.getpid() % 2000) + 23, emu, tmp), nprocs=2, join=True)


def test_grad_averager_and_shard_batch_single_process():
    _setup_paths()
    from mnk import dist as mdist
    p = torch.nn.Parameter(torch.ones(5))
    p.grad = torch.full((5,), 2.0)
    assert mdist.GradAverager([p]).average() == 0          # not initialised: no-op
    assert mdist.world_size() == 1 and mdist.rank() == 0 and not mdist.active()
    x = torch.arange(8)
    assert mdist.shard_batch(x) is x


def _bucket_worker(rank, world, port, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mnk import dist as mdist
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    unused = torch.nn.Parameter(torch.ones(4))                        # never receives a gradient
    params = list(lin.parameters()) + [unused]
    for overlap in (True, False):
        avg = mdist.GradAverager(params, bucket_mb=4e-5, overlap=overlap)    # ~10 floats per bucket: several buckets
        assert len(avg.buckets) >= 3
        for p in params:
            p.grad = None
        x = torch.full((4, 7), float(rank + 1))
        avg.arm()
        lin(x).sum().backward()
        local = [p.grad.clone() for p in lin.parameters()]
        n = avg.average()
        assert n == 6 and unused.grad is None
        # reference: plain all-reduce of the local gradients
        for p, g in zip(lin.parameters(), local):
            dist.all_reduce(g)
            assert torch.allclose(p.grad, g / world, atol=1e-6)
        for h in avg._hooks:
            h.remove()
    dist.destroy_process_group()


def test_bucketed_overlapped_gradient_averaging():
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_bucket_worker, args=(2, port, None), nprocs=2, join=True)


def _auto_worker(rank, world, port, out_dir):
    """the reference's unmodified loop shape: loss.backward(); optimizer.step() on ONE DataLoader batch that the wrapper
    scatters (replicate.py:64-67) -- averaging installed by the wrapper, the batch is rank 0's (broadcast), every rank
    computes its slice: `world` ranks == one rank on the same batch"""
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sync_batchnorm import DataParallelWithCallback

    class Full(torch.nn.Module):          # a "full model" in train.py's sense: dict batch in, per-sample losses out
        def __init__(self):
            super().__init__()
            self.net = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Tanh(), torch.nn.Linear(4, 1))

        def forward(self, x, extra=None):
            out = self.net(x["source"]) + self.net(x["video"])
            return out if extra is None else out + extra

    torch.manual_seed(0)
    full_model, ref = Full(), Full()
    ref.load_state_dict(full_model.state_dict())
    par = DataParallelWithCallback(full_model, device_ids=list(range(world)))   # a process group exists: installs the averaging
    opt = torch.optim.SGD(full_model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(3)
    batch = {"source": torch.randn(8, 6, generator=g), "video": torch.randn(8, 6, generator=g), "name": ["v%d" % i for i in range(8)]}
    # ranks > 0 hold a DIFFERENT batch (an unseeded DataLoader per process): rank 0's batch is the batch, as under DataParallel
    mine = batch if rank == 0 else {"source": torch.full((8, 6), 7.0), "video": torch.zeros(8, 6), "name": batch["name"]}
    before = {k: v.clone() for k, v in mine.items() if torch.is_tensor(v)}
    for _ in range(2):
        opt.zero_grad()
        out = par(mine)
        assert out.shape[0] == 8 // world                      # outputs stay this rank's shard
        out2 = par(mine, extra=out.detach())                   # a second wrapped call with a tensor that already is a shard
        assert out2.shape[0] == 8 // world
        (out.mean() + 0.0 * out2.mean()).backward()
        opt.step()
    for k, v in before.items():
        assert torch.equal(mine[k], v), "the wrapper must not write into its inputs"
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
    for _ in range(2):      # the whole batch on one rank
        ropt.zero_grad()
        ref(batch).mean().backward()
        ropt.step()
    for a, b in zip(full_model.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-6), (a, b)
    # evaluation wrappers (reconstruction.py:45-49) are replicas only: batch 1 is not scattered
    par.eval()
    assert par({"source": batch["source"][:1], "video": batch["video"][:1]}).shape[0] == 1
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_unmodified_loop_gets_scatter_and_gradient_averaging_from_the_wrapper(world):
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_auto_worker, args=(world, port, None), nprocs=world, join=True)


def test_wrapper_warns_about_device_ids_without_a_process_group():
    _setup_paths()
    import warnings
    from sync_batchnorm import DataParallelWithCallback
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        DataParallelWithCallback(torch.nn.Linear(2, 2), device_ids=[0, 1, 2, 3])
        DataParallelWithCallback(torch.nn.Linear(2, 2), device_ids=[0])
        DataParallelWithCallback(torch.nn.Linear(2, 2))
    assert len(w) == 1 and "torch.distributed.run" in str(w[0].message)
