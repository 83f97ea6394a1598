"""CPU: the N > 1 path over gloo with world_size 2 (RCCL on the GPU box uses the same code with backend "nccl")."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from predict_pv_yield_amd import distributed as D
    assert D.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)                      # ranks start with DIFFERENT parameters
    net = nn.Sequential(nn.Linear(6, 5), nn.Linear(5, 1))
    D.broadcast_parameters(net)
    flat0 = torch.cat([p.detach().flatten() for p in net.parameters()])
    gathered = [torch.zeros_like(flat0) for _ in range(world)]
    dist.all_gather(gathered, flat0)
    assert torch.equal(gathered[0], gathered[1])       # DDP's initial broadcast
    # rank-local gradients, then the summing all-reduce (one flat bucket + the dominant tensor in place)
    x = torch.full((4, 6), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    D.all_reduce_gradients(net)
    for p, g in zip(net.parameters(), local):
        both = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(both, g)
        assert torch.allclose(p.grad, both[0] + both[1])
    # overlapped variant: the large tensor's all-reduce is launched from an autograd hook during backward
    net2 = nn.Sequential(nn.Linear(6, 5), nn.Linear(5, 1))
    D.broadcast_parameters(net2)
    sync = D.OverlappedGradSync(net2, large_numel=20)      # the 6x5 weight counts as "large" here
    assert len(sync.large) == 1 and len(sync.small) == 3
    net2(x).sum().backward()
    assert len(sync._pending) == 1                          # launched inside backward
    local2 = None
    sync.finish()
    ref = nn.Sequential(nn.Linear(6, 5), nn.Linear(5, 1))
    ref.load_state_dict(net2.state_dict())
    tot = [torch.zeros_like(p) for p in ref.parameters()]
    for r in range(world):
        ref.zero_grad()
        ref(torch.full((4, 6), float(r + 1))).sum().backward()
        for t, p in zip(tot, ref.parameters()):
            t += p.grad
    for p, t in zip(net2.parameters(), tot):
        assert torch.allclose(p.grad, t)
    sync.remove()
    vals = D.all_reduce_mean_scalars({"MSE/Train": float(rank), "NMAE/Train": 2.0 * rank})
    assert vals == {"MSE/Train": 0.5, "NMAE/Train": 1.0}
    # row-sharded exchange of the big layer (reduce-scatter / all-gather; gloo takes the emulated forms)
    assert D.row_shard(16) == ((0, 8) if rank == 0 else (8, 16)) and D.row_shard(15) is None
    full = torch.arange(16 * 3, dtype=torch.float32).reshape(16, 3) * (rank + 1)
    shard, work = D.reduce_scatter_rows(full.clone())
    work.wait()
    r0, r1 = D.row_shard(16)
    assert torch.equal(shard, torch.arange(16 * 3, dtype=torch.float32).reshape(16, 3)[r0:r1] * 3)
    mat = torch.zeros(16, 3)
    mat[r0:r1] = float(rank + 1)            # every rank wrote only its own rows
    assert D.all_gather_rows(mat) is None   # gloo: synchronous
    assert torch.equal(mat[:8], torch.ones(8, 3)) and torch.equal(mat[8:], torch.full((8, 3), 2.0))
    lo, hi = D.shard_range(11)
    # independent units: disjoint shards, no collective -- and the SAME count on every rank (the tail item is dropped):
    # the train loop issues collectives every step, so unequal step counts would deadlock
    assert (lo, hi) == ((0, 5) if rank == 0 else (5, 10))
    for n, w in ((24900, 8), (10, 4), (10, 8), (7, 2)):
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert len({b - a for a, b in spans}) == 1 and spans[0][0] == 0 and all(s[1] == t[0] for s, t in zip(spans, spans[1:]))
        assert spans[-1][1] == (n // w) * w
    with pytest.raises(ValueError):
        D.shard_range(3, 0, 8)                                   # fewer items than ranks: hard error, not an empty shard
    # loader shards (DistributedSampler's rule): equal counts, padded by wrapping, nothing dropped, never an error
    assert D.shard_indices(5) == ([0, 2, 4] if rank == 0 else [1, 3, 0])
    for n, w in ((24900, 8), (10, 4), (2, 8), (7, 2), (1, 3)):
        parts = [D.shard_indices(n, r, w) for r in range(w)]
        assert len({len(p) for p in parts}) == 1 and len(parts[0]) == -(-n // w)
        assert set(i for p in parts for i in p) == set(range(n))
    # negotiate_grad_sync: a mode that fails on ONE rank moves EVERY rank to the next simpler mode, in-process
    class _Opt(torch.optim.SGD):
        def __init__(self, params):
            super().__init__(params, lr=0.1)
            self.large_grad_mode, self.tried = None, []

        def large_params(self):
            return []

        def set_large_grad_mode(self, mode):
            self.tried.append(mode)
            if mode == "sharded" and rank == 1:
                raise RuntimeError("simulated: reduce_scatter unsupported")
            self.large_grad_mode = mode

    class _M(nn.Linear):
        def training_step(self, batch, i):
            return (self(batch) ** 2).mean()

    m = _M(3, 1)
    D.broadcast_parameters(m)
    o = _Opt(m.parameters())
    assert D.negotiate_grad_sync(m, o, torch.ones(2, 3), "sharded") == "bf16"
    assert o.tried == ["sharded", "bf16"] and o.large_grad_mode == "bf16"
    wv = torch.cat([p.detach().flatten() for p in m.parameters()])
    both = [torch.zeros_like(wv) for _ in range(world)]
    dist.all_gather(both, wv)
    assert torch.equal(both[0], both[1])                # re-broadcast after the failed attempt, then one common step
    # ... unless the caller refuses a demotion (bench.py without --allow-demotion): EVERY rank leaves with SystemExit, also the
    # one on which the requested mode would have worked
    o2 = _Opt(m.parameters())
    try:
        D.negotiate_grad_sync(m, o2, torch.ones(2, 3), "sharded", allow_demotion=False)
        raise AssertionError("a refused demotion must end the run")
    except SystemExit as e:
        assert "'sharded' was requested but is not in force" in str(e) and o2.tried == ["sharded"]
    # the injected-failure hook (what tests/test_gpu_ddp.py uses against bench.py itself) and a mode that holds
    import os as _os
    _os.environ["PV_DIST_FAIL_MODES"] = "bf16"
    try:
        o3 = _Opt(m.parameters())
        assert D.negotiate_grad_sync(m, o3, torch.ones(2, 3), "bf16") == "autograd"
        try:
            D.negotiate_grad_sync(m, _Opt(m.parameters()), torch.ones(2, 3), "bf16", allow_demotion=False)
            raise AssertionError("a refused demotion must end the run")
        except SystemExit:
            pass
    finally:
        del _os.environ["PV_DIST_FAIL_MODES"]
    assert D.negotiate_grad_sync(m, _Opt(m.parameters()), torch.ones(2, 3), "bf16", allow_demotion=False) == "bf16"
    # the Trainer drives the same helpers: a toy fit keeps the replicas identical
    from predict_pv_yield_amd import lightning as pl

    class Toy(pl.LightningModule):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(3, 1)

        def training_step(self, batch, i):
            loss = (self.lin(batch) ** 2).mean()
            self.log_dict({"L/Train": loss}, on_step=True, on_epoch=True, sync_dist=True)
            return loss

        def configure_optimizers(self):
            return torch.optim.SGD(self.parameters(), lr=0.1)

    toy = Toy()
    data = [torch.randn(8, 3, generator=torch.Generator().manual_seed(10 * rank + i)) for i in range(3)]
    tr = pl.Trainer(gpus=0, max_epochs=1)
    tr.fit(toy, torch.utils.data.DataLoader(data, batch_size=None))
    w = toy.lin.weight.detach().flatten()
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    assert torch.allclose(ws[0], ws[1])
    metric = tr.callback_metrics["L/Train_epoch"]
    ms = [None, None]
    dist.all_gather_object(ms, metric)
    assert abs(ms[0] - ms[1]) < 1e-12                  # sync_dist: every rank logs the mean over ranks
    ret[rank] = True
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret.get(0) and ret.get(1)


def _trainer_worker(rank, world, port, tmp, ret):
    """What `python -m torch.distributed.run --nproc-per-node 2 run.py ...` gives a rank: torchrun's environment and
    nothing else.  The Trainer itself must join the process group (it did not, so every rank trained alone
    and raced on the same checkpoint files)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.dataloader import NetCDFDataModule
    assert not dist.is_initialized()

    class Toy(pl.LightningModule):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 1)
            self.seen = []

        def forward(self, batch):
            return self.lin(batch["satellite"]["data"].float().mean(dim=(1, 2, 3))[:, :4])

        def training_step(self, batch, i):
            self.seen.append(float(batch["satellite"]["data"].float().sum()))
            loss = (self(batch) ** 2).mean()
            self.log_dict({"L/Train": loss}, on_step=True, on_epoch=True, sync_dist=True)
            return loss

        def validation_step(self, batch, i):
            self.log_dict({"L/Validation": (self(batch) ** 2).mean()}, on_step=True, on_epoch=True, sync_dist=True)

        def configure_optimizers(self):
            return torch.optim.SGD(self.parameters(), lr=0.05)      # no grad_scale: the averaged all-reduce path

    if rank != 0:
        def no_write(*a, **k):
            raise AssertionError("only rank 0 may write checkpoints")
        torch.save = no_write
    torch.manual_seed(7 + rank)                                     # ranks start DIFFERENT: fit must broadcast rank 0's
    toy = Toy()
    dm = NetCDFDataModule(fake_data=True, n_train_data=5, n_val_data=2, data_path=os.path.join(tmp, "none"), batch_size=2)
    ck = pl.ModelCheckpoint(dirpath=os.path.join(tmp, "ckpt"), save_last=True, save_top_k=0)
    logger = pl.CSVLogger(save_dir=tmp, name="csv")
    tr = pl.Trainer(gpus=0, max_epochs=1, callbacks=[ck], logger=logger)
    tr.fit(toy, datamodule=dm)
    assert dist.is_initialized() and tr.world_size == 2 and tr.is_global_zero == (rank == 0)
    assert len(toy.seen) == 3                                       # 5 train batches over 2 ranks: 3 each, padded by wrapping
    w = torch.cat([p.detach().flatten() for p in toy.parameters()])
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    assert torch.equal(ws[0], ws[1])                                # replicas stayed identical
    seen = [None, None]
    dist.all_gather_object(seen, toy.seen)
    assert len(set(seen[0]) | set(seen[1])) == 5 and len(set(seen[0]) & set(seen[1])) == 1   # all 5 batches, one repeated
    ms = [None, None]
    dist.all_gather_object(ms, tr.callback_metrics["L/Train_epoch"])
    assert abs(ms[0] - ms[1]) < 1e-12
    ret[rank] = True
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_trainer_joins_the_process_group_from_torchrun_env(tmp_path):
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, port, str(tmp_path), ret), nprocs=world, join=True)
    assert ret.get(0) and ret.get(1)
    assert os.path.exists(tmp_path / "ckpt" / "last.ckpt")
    ck = torch.load(tmp_path / "ckpt" / "last.ckpt")
    assert {"state_dict", "epoch", "global_step", "optimizer_states", "pytorch-lightning_version", "callbacks",
            "lr_schedulers", "hyper_parameters"} <= set(ck)
    assert ck["epoch"] == 1 and ck["global_step"] == 3             # PL convention: the next epoch to run
    assert os.path.exists(tmp_path / "csv" / "version_0" / "metrics.csv")


def _eight_rank_worker(rank, world, port, ret):
    """config 4 runs on 8 ranks: the row-sharded exchange, the shard arithmetic and the bf16-on-the-wire sum at that width."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from predict_pv_yield_amd import distributed as D
    assert D.init_from_env(backend="gloo")
    n, k = 128, 24                                     # fc1 has 128 rows: 16 per rank
    assert D.row_shard(n) == (16 * rank, 16 * (rank + 1)) and D.row_shard(100) is None
    g = torch.Generator().manual_seed(1000 + rank)
    mine = torch.randn(n, k, generator=g)
    every = [torch.randn(n, k, generator=torch.Generator().manual_seed(1000 + r)) for r in range(world)]
    total = torch.stack(every).sum(0)                  # f32 sum in rank order == what gloo's all-reduce may reorder: allclose
    shard, work = D.reduce_scatter_rows(mine.clone())
    if work is not None:
        work.wait()
    r0, r1 = D.row_shard(n)
    assert torch.allclose(shard, total[r0:r1], rtol=1e-5, atol=1e-5)
    mat = torch.full((n, k), -1.0)
    mat[r0:r1] = float(rank)
    D.all_gather_rows(mat)
    assert torch.equal(mat, torch.arange(world, dtype=torch.float32).repeat_interleave(16)[:, None].expand(n, k))
    assert D.shard_range(64) == (8 * rank, 8 * rank + 8)
    assert D.shard_indices(2) == [rank % 2]            # the shipped experiments validate on 2 batches: no rank fails or idles
    # bf16 on the wire (HipAdam large_grad_mode "bf16" / "sharded" exchange fc1's gradient as bf16; the reference
    # all-reduces f32): 8 bf16 addends summed in bf16 stay within a few bf16 ulps of the f32 sum, norm-wise well under 1 %
    wire = mine.bfloat16()
    dist.all_reduce(wire, op=dist.ReduceOp.SUM)
    exact = torch.stack([e.bfloat16().float() for e in every]).sum(0)
    rel = ((wire.float() - exact).norm() / exact.norm()).item()
    assert rel < 6e-3, rel
    worst = ((wire.float() - exact).abs() / (torch.stack([e.abs() for e in every]).sum(0))).max().item()
    assert worst < 8 * 2.0 ** -8, worst                # each of the 7 additions rounds to 8 significant bits
    vals = D.all_reduce_mean_scalars({"NMAE/Train": float(rank)})
    assert vals == {"NMAE/Train": 3.5}
    ret[rank] = True
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_eight_rank_gloo_row_shards_and_bf16_wire_sum():
    world = 8
    port = 33500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_eight_rank_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def _ksharded_worker(rank, world, port, ret):
    """The K-sharded fc1 exchange (distributed.py, HipAdam large_grad_mode "ksharded") on CPU tensors over gloo: the four
    collectives move exactly the bytes a single process would index, and a Linear evaluated through them -- forward, input
    gradient, weight gradient of this rank's columns -- equals the dense layer on the whole global batch."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from predict_pv_yield_amd import distributed as D
    assert D.init_from_env(backend="gloo")
    b, n, k = 3, 16, 64 * world                     # per-rank batch, outputs, input features (K / W = 64)
    g = torch.Generator().manual_seed(11)
    x_all = torch.randn(world * b, k, generator=g)               # the global batch, known to every rank here
    w = torch.randn(n, k, generator=g)
    dy_all = torch.randn(world * b, n, generator=g)
    assert D.column_shard(k) == (rank * 64, (rank + 1) * 64)
    assert D.column_shard(k + 8) is None and D.column_shard(world * 12) is None      # not divisible / shard not a multiple of 8
    k0, k1 = D.column_shard(k)
    x_loc = x_all[rank * b:(rank + 1) * b].contiguous()
    # forward exchange: all samples, my columns -- the same values a single process would slice
    x_cols = D.all_to_all_columns(x_loc)
    assert x_cols.shape == (world * b, 64) and torch.equal(x_cols, x_all[:, k0:k1])
    # partial products summed in RANK ORDER for my samples
    partial = x_cols @ w[:, k0:k1].t()
    y_loc = D.reduce_scatter_sample_rows(partial)
    want = torch.zeros(b, n)
    for r in range(world):                                       # the same order, spelled out
        want += x_all[rank * b:(rank + 1) * b, r * 64:(r + 1) * 64] @ w[:, r * 64:(r + 1) * 64].t()
    assert torch.equal(y_loc, want)
    torch.testing.assert_close(y_loc, x_loc @ w.t(), rtol=1e-5, atol=1e-5)
    # backward: output gradients of all samples, then the input gradient back to its owner
    g_all = D.all_gather_sample_rows(dy_all[rank * b:(rank + 1) * b].contiguous())
    assert torch.equal(g_all, dy_all)
    dx_cols = g_all @ w[:, k0:k1]                                # [W b, K / W]: all samples, my columns
    dx_loc = D.all_to_all_rows_back(dx_cols)
    assert dx_loc.shape == (b, k) and torch.equal(dx_loc, dy_all[rank * b:(rank + 1) * b] @ w)
    # this rank's columns of the weight gradient over the WHOLE global batch = the column slice of the dense gradient
    dw_shard = g_all.t() @ x_cols
    torch.testing.assert_close(dw_shard, (dy_all.t() @ x_all)[:, k0:k1], rtol=1e-6, atol=1e-6)
    # checkpoints: column shards back into a full matrix
    full = torch.zeros(n, k)
    D.all_gather_columns(w[:, k0:k1] * 2.0, full)
    assert torch.equal(full, w * 2.0)
    ret[rank] = True
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 8])
def test_ksharded_fc1_exchange_gloo(world):
    port = 37500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ksharded_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))
