"""ORACLE (test infrastructure, not product): plain-PyTorch fp32 CPU restatement of the reference's
gesture VQ-VAE inference path — VQVAE.encode/_encode (codebook/models/vqvae.py:161-181),
Encoder/Decoder (encdec.py:53-136), ResConv1DBlock/Resnet1D (resnet.py:27-77),
BottleneckBlock.quantise/dequantise (bottleneck.py:120-130) — as a function of a checkpoint
state_dict with the reference's key names.

Pinned against the reference model itself: tests/golden/make_golden_vqvae.py instantiates the reference
VQVAE (imported from /root/reference in the build container), loads the same seeded state_dict and
records its outputs; tests/test_oracle_golden.py::test_vqvae_oracle_vs_reference compares.
"""
import numpy as np
import torch
import torch.nn.functional as F

HPS = dict(input_dim=135, width=512, emb_width=512, l_bins=512, down_t=3, stride_t=2, depth=3,
           dilation_growth_rate=3, reverse_decoder_dilation=True)


def _sd(state_dict):
    out = {}
    for k, v in state_dict.items():
        k = k[7:] if k.startswith("module.") else k
        out[k] = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).float()
    return out


def _resnet(sd, name, x, depth, growth, reverse):
    for d in range(depth):
        dil = growth ** (depth - 1 - d if reverse else d)                  # resnet.py:57-62
        h = F.conv1d(F.relu(x), sd["%s.model.%d.model.1.weight" % (name, d)],
                     sd["%s.model.%d.model.1.bias" % (name, d)], padding=dil, dilation=dil)
        h = F.conv1d(F.relu(h), sd["%s.model.%d.model.3.weight" % (name, d)],
                     sd["%s.model.%d.model.3.bias" % (name, d)])
        x = x + h                                                           # resnet.py:46
    return x


def encode_latent(state_dict, x, hps=None):
    """x: (B,T,C) float -> latent (B, E, T/8)."""
    h = dict(HPS, **(hps or {}))
    sd = _sd(state_dict)
    z = torch.as_tensor(x).float().permute(0, 2, 1)                         # vqvae.py:132-136
    enc = "encoders.0.level_blocks.0.model"
    for i in range(h["down_t"]):
        z = F.conv1d(z, sd["%s.%d.0.weight" % (enc, i)], sd["%s.%d.0.bias" % (enc, i)],
                     stride=h["stride_t"], padding=h["stride_t"] // 2)     # encdec.py:15-20
        z = _resnet(sd, "%s.%d.1" % (enc, i), z, h["depth"], h["dilation_growth_rate"], False)
    return F.conv1d(z, sd["%s.%d.weight" % (enc, h["down_t"])], sd["%s.%d.bias" % (enc, h["down_t"])], padding=1)


def quantise(state_dict, latent):
    """latent (B,E,L) -> ids (B,L) int64, plus (min, runner-up) distances for margin-aware comparisons."""
    sd = _sd(state_dict)
    k = sd["bottleneck.level_blocks.0.k"]
    B, E, L = latent.shape
    x = latent.permute(0, 2, 1).contiguous().view(-1, E)                    # bottleneck.py:96-100
    k_w = k.t()
    dist = torch.sum(x ** 2, dim=-1, keepdim=True) - 2 * torch.matmul(x, k_w) + torch.sum(k_w ** 2, dim=0, keepdim=True)
    top2 = torch.topk(dist, 2, dim=-1, largest=False).values
    ids = torch.min(dist, dim=-1)[1]
    return ids.view(B, L), top2[:, 0].view(B, L), top2[:, 1].view(B, L)


def encode(state_dict, x, hps=None):
    return quantise(state_dict, encode_latent(state_dict, x, hps))[0]


def decode(state_dict, ids, hps=None):
    """ids (B,L) int64 -> poses (B, 8L, C)."""
    h = dict(HPS, **(hps or {}))
    sd = _sd(state_dict)
    k = sd["bottleneck.level_blocks.0.k"]
    ids = torch.as_tensor(ids).long()
    z = F.embedding(ids, k).permute(0, 2, 1).contiguous()                  # bottleneck.py:128-130,145-154
    dec = "decoders.0.level_blocks.0.model"
    z = F.conv1d(z, sd[dec + ".0.weight"], sd[dec + ".0.bias"], padding=1)
    for i in range(h["down_t"]):
        z = _resnet(sd, "%s.%d.0" % (dec, i + 1), z, h["depth"], h["dilation_growth_rate"],
                    h["reverse_decoder_dilation"])
        z = F.conv_transpose1d(z, sd["%s.%d.1.weight" % (dec, i + 1)], sd["%s.%d.1.bias" % (dec, i + 1)],
                               stride=h["stride_t"], padding=h["stride_t"] // 2)      # encdec.py:45
    z = F.conv1d(z, sd["decoders.0.out.weight"], sd["decoders.0.out.bias"], padding=1)
    return z.permute(0, 2, 1)
