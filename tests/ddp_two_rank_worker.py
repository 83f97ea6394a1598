"""Worker of tests/test_gpu_ddp.py: one data-parallel rank training the reduced Conv3D model for a few steps in a given
large-gradient mode and saving the consolidated parameters of rank 0.  Either one of two ranks that SHARE the single GPU of
the test box (gloo for the collectives, PV_SINGLE_DEVICE=1; WORLD_SIZE 2 or 8, PV_TEST_GLOBAL_BATCH samples split among them), or the only rank of a one-rank RCCL group
(WORLD_SIZE=1, PV_DIST_SINGLE_RANK=1, backend "nccl": the collectives of the N > 1 path really run through RCCL).
The process group is initialised BEFORE anything touches the GPU.
Usage: python ddp_two_rank_worker.py <mode> <out.pt> <steps>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from predict_pv_yield_amd import distributed as D
from predict_pv_yield_amd.models.conv3d.model import Model
from predict_pv_yield_amd.optim import HipAdam

SMALL = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=60, history_minutes=60,
             number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=16, number_sat_channels=11,
             fc1_output_features=16, fc2_output_features=16, fc3_output_features=16)


def main():
    mode, out_path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    D.init_from_env(force=True)
    assert D.is_distributed()
    rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    dev = torch.device("cuda", D.local_device_index())
    # the reduced model's fc1 counts as the "large" layer (with 8 ranks only fc1: the other matrices' rows do not divide by 8,
    # and a large layer that cannot be row-sharded moves the whole job to the all-reduce)
    HipAdam.FUSE_MIN_NUMEL = int(os.environ.get("PV_TEST_FUSE_MIN_NUMEL", "1"))
    torch.manual_seed(518)
    model = Model(**SMALL, precision="bf16").to(dev)
    D.broadcast_parameters(model)
    opt = model.configure_optimizers()
    opt.grad_scale = 1.0 / world
    opt.set_large_grad_mode(mode)
    sync = D.OverlappedGradSync(model, large_numel=model.fc1.weight.numel())
    g = torch.Generator().manual_seed(7)
    n_global = int(os.environ.get("PV_TEST_GLOBAL_BATCH", "4"))
    sat = torch.randn(n_global, 11, 25, 16, 16, generator=g)
    pv = torch.rand(n_global, 25, 128, generator=g)
    lo, hi = D.shard_range(n_global)      # each rank trains on its contiguous share of the global batch
    batch = {"satellite": {"data": sat[lo:hi].to(dev)}, "pv": {"pv_yield": pv[lo:hi].to(dev)}}
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, 0)
        loss.backward()
        sync.finish()
        opt.step()
        losses.append(float(loss.detach()))
    y = model(batch).detach().cpu()        # forward AFTER the last step: waits for the all-gathered operand copy
    opt.consolidate_sharded()
    torch.cuda.synchronize()
    assert opt.large_grad_mode == mode, (opt.large_grad_mode, mode)
    if rank == 0:
        torch.save({"backend": torch.distributed.get_backend(), "world": world, "state": {k: v.cpu() for k, v in model.state_dict().items()}, "losses": losses, "y": y,
                    "exp_avg_fc1": opt.state[model.fc1.weight]["exp_avg"].cpu(), "mode": opt.large_grad_mode}, out_path)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
