"""Encode B=256 a few times (PMC / trace target for the conv kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import synth
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
m = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
x = torch.randn((256, 240, 135), device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    m.encode(x)
torch.cuda.synchronize()
