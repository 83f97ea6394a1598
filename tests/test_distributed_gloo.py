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
    assert (lo, hi) == ((0, 6) if rank == 0 else (6, 11))      # independent units: disjoint shards, no collective
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
