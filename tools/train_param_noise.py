"""Fraction of sampled weights whose Adam update differs from the reference golden's by more than 1e-6 after training
steps 1 and 2 (the noise-floor criterion of tests/test_gpu_vqvae_train.py), worst parameters first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import test_gpu_vqvae_train as T
from qpgesture_amd.optim import Adam
from tests.golden.make_golden_vqvae import TRAIN_BETAS, TRAIN_LR, TRAIN_SEED
g = T.load_golden("vqvae_train_w512_s7")
m, sd, torch = T._setup()
x = T._x(int(g["meta"][2])).cuda()
names = [str(n) for n in g["param_names"]]
m.train()
opt = Adam(m.parameters(), lr=TRAIN_LR, betas=TRAIN_BETAS)
torch.manual_seed(TRAIN_SEED)
for step in (1, 2):
    tag = "step%d" % step
    opt.zero_grad()
    m(x); m.backward()
    grads = m.named_gradients()
    gerr = []
    for i, n in enumerate(names):
        rms = float(g[tag + "_grad_norm"][i]) / np.sqrt(grads[n].numel())
        gerr.append(float(np.abs(grads[n].numpy().reshape(-1)[::T.SUB] - g["%s_grad_%03d" % (tag, i)]).max() / rms))
    opt.step()
    params = m.state_dict()
    rows = []
    for i, n in enumerate(names):
        d = np.abs(params[n].numpy().reshape(-1)[::T.SUB] - g["%s_param_%03d" % (tag, i)])
        rows.append((float(np.mean(d > 1e-6)), float(d.max()), d.size, gerr[i], n))
    rows.sort(reverse=True)
    print("step", step, "worst grad err / rms %.3f" % max(gerr))
    for r in rows[:5]:
        print("  frac %.4f  max %.2e  samples %d  grad err/rms %.3f  %s" % r)
