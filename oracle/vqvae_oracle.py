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


# ----------------------------------------------------------------------------------------------
# VQVAE.forward (vqvae.py:183-302) and BottleneckBlock.forward / update_k (bottleneck.py:63-94, 156-186)
# ----------------------------------------------------------------------------------------------
def _l1(a, b):
    return torch.mean(torch.abs(b - a))                                     # vqvae.py:49-50  _loss_fn(x_target, x_pred)


def losses(x_target, x_out, commit_loss, hps_commit=0.02, vel=1.0, acc=1.0, reg=0.0):
    """The loss terms of vqvae.py:244-267 for levels == 1."""
    x_target = x_target.float()
    recons = _l1(x_target, x_out)
    regularization = torch.mean((x_out[:, 2:] + x_out[:, :-2] - 2 * x_out[:, 1:-1]) ** 2)
    velocity = _l1(x_out[:, 1:] - x_out[:, :-1], x_target[:, 1:] - x_target[:, :-1])
    acceleration = _l1(x_out[:, 2:] + x_out[:, :-2] - 2 * x_out[:, 1:-1],
                       x_target[:, 2:] + x_target[:, :-2] - 2 * x_target[:, 1:-1])
    loss = recons + commit_loss * hps_commit + reg * regularization + vel * velocity + acc * acceleration
    return loss, dict(recons_loss_l1=recons, recons_loss=recons, l1_loss=recons, commit_loss=commit_loss,
                      regularization=regularization, velocity_loss=velocity, acceleration_loss=acceleration)


def ema_update(x, x_l, k, k_sum, k_elem, k_rand, mu=0.99, threshold=1.0):
    """BottleneckBlock.update_k (bottleneck.py:63-94) given the random-restart rows `k_rand` (the reference draws
    them with y[t.randperm(n)][:k_bins]).  x (R,E), x_l (R,) int64.  Returns new (k, k_sum, k_elem) and the metrics."""
    k_bins, E = k.shape
    onehot = torch.zeros(k_bins, x.shape[0])
    onehot.scatter_(0, x_l.view(1, -1), 1)
    _k_sum = torch.matmul(onehot, x)
    _k_elem = onehot.sum(dim=-1)
    old_k = k
    k_sum = mu * k_sum + (1. - mu) * _k_sum
    k_elem = mu * k_elem + (1. - mu) * _k_elem
    usage = (k_elem.view(k_bins, 1) >= threshold).float()
    k_new = usage * (k_sum.view(k_bins, E) / k_elem.view(k_bins, 1)) + (1 - usage) * k_rand
    _k_prob = _k_elem / torch.sum(_k_elem)
    entropy = -torch.sum(_k_prob * torch.log(_k_prob + 1e-8))
    used_curr = (_k_elem >= threshold).sum()
    dk = torch.norm(k_new - old_k) / np.sqrt(np.prod(old_k.shape))
    return k_new, k_sum, k_elem, dict(entropy=entropy, used_curr=used_curr, usage=torch.sum(usage), dk=dk)


def forward(state_dict, x, hps=None, training=False, ema_state=None, generator_seed=None, commit=0.02, vel=1.0,
            acc=1.0, reg=0.0):
    """VQVAE.forward for levels == 1.  Returns (x_out (B,T,C), loss, metrics, new_ema_state).
    training=True also runs the EMA codebook update; `ema_state` = dict(k, k_sum, k_elem, init) carried between
    steps (None = a freshly constructed BottleneckBlock: init False).  Random draws (init_k / random restart) use
    torch.randperm on the global CPU generator exactly like the reference (bottleneck.py:43,72)."""
    sd = _sd(state_dict)
    x = torch.as_tensor(x).float()
    if generator_seed is not None:
        torch.manual_seed(generator_seed)
    lat = encode_latent(state_dict, x, hps)                                  # (B,E,L)
    B, E, L = lat.shape
    z = lat.permute(0, 2, 1).contiguous().view(-1, E)                       # bottleneck.py:96-100
    prenorm = torch.norm(z - torch.mean(z)) / np.sqrt(np.prod(z.shape))
    st = dict(ema_state) if ema_state is not None else dict(k=sd["bottleneck.level_blocks.0.k"], k_sum=None,
                                                           k_elem=None, init=False)
    if training and not st["init"]:                                          # init_k (bottleneck.py:39-49), R >= k_bins
        kb = st["k"].shape[0]
        assert z.shape[0] >= kb, "oracle covers the untiled case (rows >= k_bins)"
        st["k"] = z[torch.randperm(z.shape[0])][:kb].detach().clone()
        st["k_sum"] = st["k"].clone()
        st["k_elem"] = torch.ones(kb)
        st["init"] = True
    k = st["k"]
    k_w = k.t()
    dist = torch.sum(z ** 2, dim=-1, keepdim=True) - 2 * torch.matmul(z, k_w) + torch.sum(k_w ** 2, dim=0, keepdim=True)
    min_d, x_l = torch.min(dist, dim=-1)
    fit = torch.mean(min_d)
    x_d = F.embedding(x_l, k)
    qm = {}
    if training:
        kb = k.shape[0]
        k_rand = z[torch.randperm(z.shape[0])][:kb].detach()                 # bottleneck.py:71-72
        k_new, st["k_sum"], st["k_elem"], um = ema_update(z.detach(), x_l, k, st["k_sum"], st["k_elem"], k_rand)
        st["k"] = k_new
        qm = dict(fit=fit, pn=prenorm, **um)
        # average_metrics floors with `//` (models/utils/logger.py:50)
        qm = {kk: torch.as_tensor(v).float() // 1 for kk, v in qm.items()}
    commit_loss = torch.norm(x_d.detach() - z) ** 2 / np.prod(z.shape)
    x_d = z + (x_d - z).detach()                                             # straight-through (bottleneck.py:179)
    zq = x_d.view(B, L, E).permute(0, 2, 1)
    x_out = decode_from_latent(state_dict, zq, hps)
    loss, metrics = losses(x, x_out, commit_loss, hps_commit=commit, vel=vel, acc=acc, reg=reg)
    metrics.update(qm)
    return x_out, loss, metrics, st, x_l.view(B, L)


def decode_from_latent(state_dict, zq, hps=None):
    """Decoder.forward on a quantised latent (B,E,L) -> (B,8L,C)."""
    h = dict(HPS, **(hps or {}))
    sd = _sd(state_dict)
    dec = "decoders.0.level_blocks.0.model"
    z = F.conv1d(zq, sd[dec + ".0.weight"], sd[dec + ".0.bias"], padding=1)
    for i in range(h["down_t"]):
        z = _resnet(sd, "%s.%d.0" % (dec, i + 1), z, h["depth"], h["dilation_growth_rate"],
                    h["reverse_decoder_dilation"])
        z = F.conv_transpose1d(z, sd["%s.%d.1.weight" % (dec, i + 1)], sd["%s.%d.1.bias" % (dec, i + 1)],
                               stride=h["stride_t"], padding=h["stride_t"] // 2)
    z = F.conv1d(z, sd["decoders.0.out.weight"], sd["decoders.0.out.bias"], padding=1)
    return z.permute(0, 2, 1)
