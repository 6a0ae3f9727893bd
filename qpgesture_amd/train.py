#!/usr/bin/env python
"""Gesture VQ-VAE training on MI355X — the loop of codebook/train.py:53-153 with every tensor operation on HIP
kernels (qpgesture_amd.vqvae.VQVAE forward / backward, qpgesture_amd.optim.Adam):

    python -m qpgesture_amd.train --config qpgesture_amd/configs/codebook.yml --gpu 0 \\
        --train_data train_poses.npz --val_data val_poses.npz

Per epoch, like the reference: evaluate on the validation set (mean joint-wise Euclidean error, train.py:27-51),
save `<name>_checkpoint_best.bin` / `<name>_checkpoint_<epoch>.bin` = {'args', 'epoch', 'model_dict'} with the
DataParallel `module.` names (readable by the reference's VisualizeCodebook.py / make_beat_dataset.py), then one
pass over the shuffled training windows: zero_grad, forward, backward, Adam step; MultiStepLR per epoch.

Data: the reference reads pose windows from an lmdb cache through pyarrow.deserialize (data_loader/lmdb_data_
loader.py:48-74); neither package is in this image, so windows come from `.npz` (`poses` (N, n_poses, 135)) / `.npy`
files of UN-normalised poses and are normalised with the config's data_mean / data_std exactly as the dataset class
does (:65-66).  `--synthetic N` trains on N seeded random windows (smoke runs, benchmarks).

Multi-GPU: one process per GPU under torchrun (`python -m torch.distributed.run --nproc-per-node N -m
qpgesture_amd.train ...`).  Each rank takes its slice of every global batch; gradients are averaged with ONE
all-reduce of the flat gradient buffer per step, and the codebook's batch statistics / random restarts use the
collectives of bottleneck.py:44,73-75, so every rank holds identical weights and codebook."""
import argparse
import logging
import os
import time

import numpy as np
import torch

from . import parallel
from .checkpoint import load_config
from .optim import Adam, MultiStepLR
from .vqvae import ActivationRange, VQVAE, init_state_dict, normalize_poses


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Codebook")                       # configs/parse_args.py
    p.add_argument("--config", default=os.path.join(os.path.dirname(__file__), "configs", "codebook.yml"))
    p.add_argument("--gpu", type=str, default="0")
    p.add_argument("--train_data", default=None, help=".npz/.npy pose windows (overrides train_data_path)")
    p.add_argument("--val_data", default=None)
    p.add_argument("--synthetic", type=int, default=0, help="train on N seeded random windows instead of files")
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--model_save_path", default=None)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--resume", default=None, help="checkpoint to start from")
    p.add_argument("--max_updates", type=int, default=0, help="stop after this many optimiser steps (0 = all)")
    p.add_argument("--train_precision", choices=["f32", "f16x3"], default="f32",
                   help="forward convolutions of a training step: f32 matrix cores (default; the reference's arithmetic class) "
                        "or split-operand f16 (three f16 MFMAs per product, f32 accumulation: ~1e-5 of the f32 kernels, half "
                        "the forward time; the backward pass is f32 either way; a step whose activations leave the f16 range "
                        "is redone in f32 and the run stays in f32)")
    return p.parse_args(argv)


def load_windows(path, data_mean, data_std):
    a = np.load(path, allow_pickle=False)
    poses = a["poses"] if hasattr(a, "files") else a
    poses = np.asarray(poses, np.float64).reshape(poses.shape[0], poses.shape[1], -1)
    flat = normalize_poses(poses.reshape(-1, poses.shape[-1]), data_mean, data_std)
    return torch.from_numpy(flat.reshape(poses.shape).astype(np.float32))


def evaluate_testset(model, windows, batch_size):
    """train.py:27-51: per batch mean over (b, t, joint) of the 9-channel Euclidean error; returns (mean, std) over
    batches (drop_last like the reference's loader).  The arithmetic on the decoded poses is a 4-op epilogue."""
    model.eval()
    errs = []
    for i in range(0, windows.shape[0] - batch_size + 1, batch_size):
        x = windows[i:i + batch_size].to(model.device)
        out, _, _ = model(x)
        b, t, c = x.shape
        diff = (x - out).view(b, t, c // 9, 9)
        errs.append(torch.mean(torch.sqrt(torch.sum(diff ** 2, dim=3))))
    model.train()
    if not errs:
        return float("nan"), float("nan")
    e = torch.stack(errs)
    return float(e.mean()), float(e.std()) if e.numel() > 1 else 0.0


def main(argv=None):
    args = parse_args(argv)
    logging.getLogger().setLevel(logging.INFO)
    cfg = load_config(args.config)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device("cuda:%d" % (local if world > 1 and not os.environ.get("QPG_TRAIN_ONE_GPU") else int(args.gpu)))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("QPG_DIST_BACKEND", "nccl"))   # "nccl" is RCCL on ROCm
    seed = args.seed if args.seed is not None else 0
    torch.manual_seed(seed)                                      # same init + same restart draws on every rank

    epochs = args.epochs if args.epochs is not None else cfg.epochs
    batch = args.batch_size if args.batch_size is not None else cfg.batch_size
    save_dir = args.model_save_path or cfg.model_save_path
    assert batch % world == 0, "batch_size must divide over the ranks"
    if args.synthetic:
        g = np.random.Generator(np.random.PCG64(seed))
        train = torch.from_numpy(g.standard_normal((args.synthetic, cfg.n_poses, 135)).astype(np.float32))
        val = torch.from_numpy(g.standard_normal((max(batch, 1), cfg.n_poses, 135)).astype(np.float32))
    else:
        train = load_windows(args.train_data or cfg.train_data_path, cfg.data_mean, cfg.data_std)
        val = load_windows(args.val_data or cfg.val_data_path, cfg.data_mean, cfg.data_std)
    logging.info("train windows: %d, validation windows: %d", train.shape[0], val.shape[0])

    model = VQVAE(cfg.VQVAE, 15 * 9, device=dev)
    model.train_precision = args.train_precision
    start_epoch = 1
    if args.resume:
        from .checkpoint import load_checkpoint
        ck = load_checkpoint(args.resume)
        model.load_state_dict(ck["model_dict"])
        model.k_init = bool(ck.get("k_init", True))
        if model.k_init:
            model.k_sum = ck["k_sum"].to(dev) if "k_sum" in ck else model.k.clone()
            model.k_elem = ck["k_elem"].to(dev) if "k_elem" in ck else torch.ones(model.bins, device=dev)
        start_epoch = int(ck.get("epoch") or 0) + 1
    else:
        ck = None
        model.load_state_dict(init_state_dict(cfg.VQVAE, 15 * 9, seed=seed))
    opt = Adam(model.parameters(), lr=cfg.lr, betas=cfg.betas)
    sched = MultiStepLR(opt, milestones=cfg.milestones, gamma=cfg.gamma)
    for _ in range(1, start_epoch):
        sched.step()
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    best = (1e2, 0)
    if ck is not None:
        # additive checkpoint keys of this implementation (a reference checkpoint has none of them: weights-only resume)
        if ck.get("optimizer") is not None:
            lr_now = opt.lr
            opt.load_state_dict(ck["optimizer"])
            opt.lr = lr_now                                      # the schedule above is authoritative
        if ck.get("best") is not None:
            best = (float(ck["best"][0]), int(ck["best"][1]))
    updates = 0
    per_rank = batch // world
    n_batches = train.shape[0] // batch                          # drop_last=True (train.py:62)
    for epoch in range(start_epoch, epochs + 1):
        logging.info("Epoch: %d", epoch)
        mean, std = evaluate_testset(model, val, min(batch, val.shape[0]))
        logging.info("diff mean on validation: %.3f, diff std on validation: %.3f", mean, std)
        is_best = mean < best[0]
        if is_best:
            logging.info(" *** BEST VALIDATION LOSS : %.3f", mean)
            best = (mean, epoch)
        else:
            logging.info(" best validation loss so far: %.3f at EPOCH %d", best[0], best[1])
        if rank == 0 and (is_best or epoch % cfg.save_per_epochs == 0):
            name = ("%s/%s_checkpoint_best.bin" if is_best else "%s/%s_checkpoint_%03d.bin")
            name = name % ((save_dir, cfg.name) if is_best else (save_dir, cfg.name, epoch))
            # 'args' as PLAIN nested dicts / lists (no class of this package inside the pickle), so that the reference's
            # own VisualizeCodebook / make_beat_dataset can torch.load the file; optimizer moments and the best-so-far
            # record ride along for --resume
            torch.save({"args": _plain(cfg), "epoch": epoch, "model_dict": model.state_dict(prefix="module."),
                        "k_init": model.k_init, "k_sum": None if model.k_sum is None else model.k_sum.cpu(),
                        "k_elem": None if model.k_elem is None else model.k_elem.cpu(),
                        "optimizer": opt.state_dict(), "best": [float(best[0]), int(best[1])]}, name)
            logging.info("Saved the checkpoint")
        model.train()
        perm = torch.randperm(train.shape[0])                    # shuffle=True; identical on every rank (same seed)
        t0 = time.time()
        for i in range(n_batches):
            idx = perm[i * batch + rank * per_rank: i * batch + (rank + 1) * per_rank]
            x = train[idx].to(dev, non_blocking=True)
            opt.zero_grad()
            try:
                _, loss, metrics = model(x)
                model.backward(sync_grads=True)      # decoder-half all-reduce overlaps the encoder half of backward
            except ActivationRange as e:             # (split-f16 forward only; the status word is MAX-reduced over the
                logging.warning("%s", e)             # ranks before it is read: every rank redoes the step)
                model.train_precision = "f32"
                opt.zero_grad()
                _, loss, metrics = model(x)
                model.backward(sync_grads=True)
            opt.step()
            updates += 1
            if rank == 0:
                eta = (time.time() - t0) / (i + 1) * (n_batches - i - 1)
                logging.info("> epoch [%d] updates[%d] updates[%d] loss[%.8f] eta[%ds]", epoch, i + 1, updates - 1,
                             float(loss), int(eta))
            if args.max_updates and updates >= args.max_updates:
                break
        sched.step()
        if args.max_updates and updates >= args.max_updates:
            break
    logging.info("--------- Final best loss values ---------")
    logging.info("diff mean: %.3f at EPOCH %d", best[0], best[1])
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return model


def _plain(x):
    """AttrDict / tuples / NumPy scalars -> plain dict / list / Python scalars, recursively."""
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, np.generic):
        return x.item()
    if isinstance(x, np.ndarray):
        return x.tolist()
    return x


if __name__ == "__main__":
    main()
