"""Worker of test_gpu_vqvae_train.py::test_data_parallel_step (launched with torch.distributed.run, 2 ranks sharing
cuda:0, gloo transport): the data-parallel training step of qpgesture_amd.train vs the same step on the full batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qpgesture_amd import parallel, synth  # noqa: E402
from qpgesture_amd.optim import Adam  # noqa: E402
from qpgesture_amd.vqvae import VQVAE  # noqa: E402


def model(hps, mu):
    sd = synth.make_vqvae_state_dict(7, hps)
    m = VQVAE(dict(hps, vel=1, acc=1, commit=0.02, l_mu=mu), input_dim=135, device="cuda:0").load_state_dict(sd)
    m.train()
    m.k_init = True
    m.k_sum, m.k_elem = m.k.clone(), torch.ones(hps["l_bins"], device="cuda:0")
    return m


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hps = dict(width=64, emb_width=64, l_bins=96)
    per = 3
    x_all = torch.from_numpy(np.random.Generator(np.random.PCG64(8)).standard_normal((per * world, 96, 135))
                             .astype(np.float32)).cuda()
    # (1) frozen codebook (mu = 1): averaged shard gradients == full-batch gradients
    m = model(hps, 1.0)
    _, loss, _ = m(x_all[rank * per:(rank + 1) * per])
    m.backward()
    parallel.allreduce_sum_(m.grad, average=True)
    loss_mean = loss.clone()
    parallel.allreduce_sum_(loss_mean, average=True)
    ref = model(hps, 1.0)
    _, loss_full, _ = ref(x_all)
    ref.backward()
    err = float((m.grad - ref.grad).abs().max())
    scale = float(ref.grad.abs().max())
    assert err <= 1e-5 * scale + 1e-9, (err, scale)                 # every sample's forward is bit-identical on both sides
    # commit / recons terms are means over equal shards; the velocity/acceleration means too
    assert abs(float(loss_mean) - float(loss_full)) < 1e-5 * abs(float(loss_full)), (float(loss_mean), float(loss_full))

    # (2) live codebook (EMA + random restarts): two optimiser steps keep the ranks bit-identical
    torch.manual_seed(5)
    m = model(hps, 0.99)
    opt = Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    for step in range(2):
        _, loss, met = m(x_all[rank * per:(rank + 1) * per])
        m.backward(sync_grads=True)
        opt.step()
    sig = torch.stack([m.param.double().sum(), m.param.double().abs().sum(), m.k.double().sum(),
                       m.k_elem.double().sum(), m.k_sum.double().abs().sum()]).cpu()
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(sigs, sig)
    for s in sigs[1:]:
        assert torch.equal(s, sigs[0]), (sigs,)
    assert float(met["used_curr"]) <= per * world * 12              # codes hit by the GLOBAL batch (all-reduced counts)
    dist.barrier()
    if rank == 0:
        print("DIST_TRAIN_OK", err, scale)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
