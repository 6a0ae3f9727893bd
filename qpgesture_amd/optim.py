"""Optimiser pieces of the reference's codebook training loop (codebook/train.py:71-72, 130, 148):
`optim.Adam(model.parameters(), lr, betas)` and `lr_scheduler.MultiStepLR(milestones, gamma)`, over the model's
flat parameter / gradient buffers (qpgesture_amd.vqvae.VQVAE.parameters()).  One HIP launch per step."""
import torch

from . import _lib


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.param, self.grad = params
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(self.param)
        self.exp_avg_sq = torch.zeros_like(self.param)
        self.steps = 0

    def zero_grad(self):
        """backward() overwrites every gradient; kept for the reference loop's shape (train.py:123)."""

    def step(self):
        self.steps += 1
        _lib.call("qpg_adam_step_f32", self.param.device, self.param, self.grad, self.exp_avg, self.exp_avg_sq,
                  self.param.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.steps)

    def state_dict(self):
        return dict(lr=self.lr, betas=self.betas, eps=self.eps, steps=self.steps, exp_avg=self.exp_avg.cpu(),
                    exp_avg_sq=self.exp_avg_sq.cpu())

    def load_state_dict(self, sd):
        self.lr, self.betas, self.eps, self.steps = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["steps"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class MultiStepLR:
    """lr = base_lr * gamma ** (number of milestones <= epoch), stepped once per epoch (train.py:148)."""

    def __init__(self, optimizer, milestones, gamma=0.1):
        self.opt, self.milestones, self.gamma = optimizer, sorted(int(m) for m in milestones), float(gamma)
        self.base_lr, self.epoch = optimizer.lr, 0

    def step(self):
        self.epoch += 1
        self.opt.lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.epoch)
