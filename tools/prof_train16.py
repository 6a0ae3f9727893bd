import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from qpgesture_amd import synth
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
m = VQVAE(dict(vel=1, acc=1), 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7)).train()
m.train_precision = os.environ.get("PREC", "f16x3")
x = torch.randn((256, 240, 135), device=dev)
for _ in range(8): m(x)
torch.cuda.synchronize()
