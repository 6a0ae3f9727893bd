"""Gesture VQ-VAE inference on the MI355X: mirrors the reference's `VQVAE.encode` / `VQVAE.decode`
(codebook/models/vqvae.py:152-181) over the C ABI (qpg_conv1d_f32, qpg_vq_argmin_f32, qpg_vq_gather_f32).

    model = VQVAE(hps, input_dim=135, device="cuda:0")
    model.load_state_dict(torch.load(ckpt, map_location="cpu")["model_dict"])   # keys may carry `module.`
    ids   = model.encode(x)[0]          # x (B,T,135) float -> LongTensor (B,T/8)      (vqvae.py:174-181)
    poses = model.decode([ids])         # LongTensor (B,L) -> FloatTensor (B,8L,135)   (vqvae.py:152-159)

Python only repacks the weights once (Conv1d (Cout,Cin,k) -> [k][Cin_pad][Cout_pad]; ConvTranspose1d
(Cin,Cout,4) -> two 2-tap sets, one per output parity) and issues the layer sequence; every layer is a
HIP kernel launch.  No torch.nn / cuDNN / MIOpen call, no CPU fallback.
"""
import numpy as np
import torch

from . import _lib

DEFAULT_HPS = dict(width=512, emb_width=512, l_bins=512, downs_t=[3], strides_t=[2], depth=3, m_conv=1.0,
                   dilation_growth_rate=3, vqvae_reverse_decoder_dilation=True, levels=1)

BK, BN = 16, 128          # K / N padding the conv kernel's tile needs (csrc/qpg_vqvae.hip)


def _get(hps, k):
    if isinstance(hps, dict):
        return hps.get(k, DEFAULT_HPS[k])
    return getattr(hps, k, DEFAULT_HPS[k])


def _pad(n, m):
    return (n + m - 1) // m * m


class _Conv:
    """One packed convolution: weights [taps][Cin_pad][Cout_pad], bias [Cout_pad]."""

    def __init__(self, w_tap_ci_co, bias, dev):
        taps, cin, cout = w_tap_ci_co.shape
        self.taps, self.cin, self.cout = taps, cin, cout
        self.cin_pad, self.cout_pad = _pad(cin, BK), _pad(cout, BN)
        w = torch.zeros((taps, self.cin_pad, self.cout_pad), dtype=torch.float32)
        w[:, :cin, :cout] = w_tap_ci_co
        b = torch.zeros((self.cout_pad,), dtype=torch.float32)
        b[:cout] = bias
        self.w, self.b = w.to(dev).contiguous(), b.to(dev).contiguous()


class VQVAE:
    def __init__(self, hps=None, input_dim=135, device="cuda:0"):
        hps = hps or {}
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("VQVAE needs a HIP device (got %s); there is no CPU path" % dev)
        _lib.load()
        self.device = dev
        self.input_dim = input_dim
        self.width, self.emb = _get(hps, "width"), _get(hps, "emb_width")
        self.bins = _get(hps, "l_bins")
        if _get(hps, "levels") != 1:
            raise NotImplementedError("levels != 1 (codebook.yml uses 1)")
        self.down_t, self.stride_t = _get(hps, "downs_t")[0], _get(hps, "strides_t")[0]
        if self.stride_t != 2:
            raise NotImplementedError("stride_t != 2 (codebook.yml uses 2)")
        self.depth = _get(hps, "depth")
        self.growth = _get(hps, "dilation_growth_rate")
        self.reverse = bool(_get(hps, "vqvae_reverse_decoder_dilation"))
        self.hop = self.stride_t ** self.down_t
        self._loaded = False

    # ------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """Accepts the reference checkpoint's `model_dict` (train.py:114-116; keys with or without the
        DataParallel `module.` prefix) and repacks it for the kernels."""
        sd = {}
        for k, v in state_dict.items():
            k = k[7:] if k.startswith("module.") else k
            sd[k] = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).detach().float().cpu()
        dev = self.device

        def conv(name):               # nn.Conv1d weight (Cout, Cin, k) -> [k][Cin][Cout]
            return _Conv(sd[name + ".weight"].permute(2, 1, 0).contiguous(), sd[name + ".bias"], dev)

        def resnet(name):
            return [(conv("%s.model.%d.model.1" % (name, d)), conv("%s.model.%d.model.3" % (name, d)))
                    for d in range(self.depth)]

        enc = "encoders.0.level_blocks.0.model"
        self.enc_down = [(conv("%s.%d.0" % (enc, i)), resnet("%s.%d.1" % (enc, i))) for i in range(self.down_t)]
        self.enc_out = conv("%s.%d" % (enc, self.down_t))
        dec = "decoders.0.level_blocks.0.model"
        self.dec_in = conv(dec + ".0")
        self.dec_up = []
        for i in range(self.down_t):
            wt = sd["%s.%d.1.weight" % (dec, i + 1)]            # ConvTranspose1d: (Cin, Cout, 4)
            bt = sd["%s.%d.1.bias" % (dec, i + 1)]
            # y[2m]   = x[m-1].W[:,:,3] + x[m].W[:,:,1]        (t = 2i - 1 + k, encdec.py:45: k4 s2 p1)
            # y[2m+1] = x[m].W[:,:,2]   + x[m+1].W[:,:,0]
            even = _Conv(torch.stack((wt[:, :, 3], wt[:, :, 1])).contiguous(), bt, dev)
            odd = _Conv(torch.stack((wt[:, :, 2], wt[:, :, 0])).contiguous(), bt, dev)
            self.dec_up.append((resnet("%s.%d.0" % (dec, i + 1)), even, odd))
        self.dec_out = conv("decoders.0.out")
        k = sd["bottleneck.level_blocks.0.k"]                   # (bins, emb)
        self.k = k.to(dev).contiguous()
        self.kT = _Conv(k.t().contiguous()[None], torch.zeros(self.bins), dev)      # x.k^T as a 1-tap "conv"
        self.kk = torch.sum(k.t() ** 2, dim=0).to(dev).contiguous()                # bottleneck.py:123
        self._desc = self._build_descriptor()
        # split-K scratch of the per-layer path (the whole-network calls carve theirs out of the workspace)
        self._split_ws = torch.empty((8 * 2048 * _pad(max(self.width, self.emb, self.bins), BN),), dtype=torch.float32,
                                     device=dev)
        self._loaded = True
        return self

    def _build_descriptor(self):
        """qpg_vq_model (include/qpg.h): pointers into the packed tensors kept alive by this object."""
        m = _lib.VqModel()
        m.in_dim, m.width, m.emb, m.bins = self.input_dim, self.width, self.emb, self.bins
        m.down_t, m.depth, m.growth, m.reverse_dec = self.down_t, self.depth, self.growth, int(self.reverse)
        if self.down_t > _lib.QPG_VQ_MAX_DOWN or self.depth > _lib.QPG_VQ_MAX_DEPTH:
            raise NotImplementedError("down_t/depth beyond the descriptor's capacity")

        def fill(d, c):
            d.w, d.b = c.w.data_ptr(), c.b.data_ptr()
            d.taps, d.cin, d.cin_pad, d.cout, d.cout_pad = c.taps, c.cin, c.cin_pad, c.cout, c.cout_pad
        for i, (c, res) in enumerate(self.enc_down):
            fill(m.enc_down[i], c)
            for d, (c3, c1) in enumerate(res):
                fill(m.enc_res[i][d][0], c3)
                fill(m.enc_res[i][d][1], c1)
        fill(m.enc_out, self.enc_out)
        fill(m.dec_in, self.dec_in)
        for i, (res, even, odd) in enumerate(self.dec_up):
            for d, (c3, c1) in enumerate(res):
                fill(m.dec_res[i][d][0], c3)
                fill(m.dec_res[i][d][1], c1)
            fill(m.dec_up_even[i], even)
            fill(m.dec_up_odd[i], odd)
        fill(m.dec_out, self.dec_out)
        fill(m.kT, self.kT)
        m.k, m.kk = self.k.data_ptr(), self.kk.data_ptr()
        return m

    def _workspace(self, B, T):
        import ctypes
        n = _lib.load().qpg_vq_workspace_floats(ctypes.byref(self._desc), B, T)
        if n < 0:
            raise RuntimeError("qpg_vq_workspace_floats failed")
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < n:
            self._ws = ws = torch.empty((n,), dtype=torch.float32, device=self.device)
        return ws

    def encode_fused(self, x, return_latent=False, return_margin=False):
        """One C call for the whole encoder + quantiser (qpg_vq_encode_f32)."""
        assert self._loaded, "load_state_dict first"
        x = x.to(self.device, torch.float32).contiguous()
        B, T, _ = x.shape
        L = T // self.hop
        ws = self._workspace(B, T)
        ids = torch.empty((B, L), dtype=torch.int64, device=self.device)
        lat = torch.empty((B, L, self.emb), dtype=torch.float32, device=self.device) if return_latent else None
        mar = torch.empty((B, L), dtype=torch.float32, device=self.device) if return_margin else None
        _lib.call("qpg_vq_encode_f32", self.device, self._desc, x, B, T, ws, ws.numel(), ids, lat, mar)
        out = (ids,)
        if return_latent:
            out += (lat,)
        if return_margin:
            out += (mar,)
        return out if len(out) > 1 else ids

    # ------------------------------------------------------------------------------------------
    def _conv(self, c, x, B, T_in, T_out, in_stride=1, in_offset=0, dil=1, out=None, out_stride=1, out_offset=0,
              T_y=None, residual=None, relu_in=False, relu_out=False):
        T_y = T_out if T_y is None else T_y
        if out is None:
            out = torch.empty((B, T_y, c.cout), dtype=torch.float32, device=self.device)
        _lib.call("qpg_conv1d_f32", self.device, x, B, T_in, c.cin, c.w, c.b, c.taps, c.cin_pad, c.cout, c.cout_pad,
                  in_stride, in_offset, dil, T_out, out_stride, out_offset, T_y, residual, int(relu_in), int(relu_out),
                  out, self._split_ws, self._split_ws.numel())
        return out

    def _resnet(self, blocks, x, B, T, reverse):
        for d, (c3, c1) in enumerate(blocks):
            dil = self.growth ** (self.depth - 1 - d if reverse else d)              # resnet.py:57-62
            h = self._conv(c3, x, B, T, T, in_offset=-dil, dil=dil, relu_in=True, relu_out=True)
            x = self._conv(c1, h, B, T, T, residual=x)                                # x + conv1(relu(conv3(relu(x))))
        return x

    def encode_latent(self, x):
        """(B,T,C) float tensor on the device -> channels-last latent (B, T/8, emb)."""
        assert self._loaded, "load_state_dict first"
        x = x.to(self.device, torch.float32).contiguous()
        B, T, _ = x.shape
        for c, res in self.enc_down:
            T_out = T // self.stride_t
            x = self._conv(c, x, B, T, T_out, in_stride=self.stride_t, in_offset=-(self.stride_t // 2))
            T = T_out
            x = self._resnet(res, x, B, T, False)
        return self._conv(self.enc_out, x, B, T, T, in_offset=-1)

    def quantise(self, z, return_margin=False):
        """BottleneckBlock.quantise on a channels-last latent (B,L,emb) -> ids (B,L) int64."""
        B, L, E = z.shape
        R = B * L
        z2 = z.contiguous().view(1, R, E)
        dot = self._conv(self.kT, z2, 1, R, R)
        ids = torch.empty((R,), dtype=torch.int64, device=self.device)
        dmin = torch.empty((R,), dtype=torch.float32, device=self.device) if return_margin else None
        dsec = torch.empty((R,), dtype=torch.float32, device=self.device) if return_margin else None
        _lib.call("qpg_vq_argmin_f32", self.device, z2, dot, self.kk, R, E, self.bins, ids, dmin, dsec)
        if return_margin:
            return ids.view(B, L), (dsec - dmin).view(B, L)
        return ids.view(B, L)

    def encode(self, x, start_level=0, end_level=None, bs_chunks=1):
        """VQVAE.encode (vqvae.py:174-181): returns [LongTensor (B, T/8)]."""
        x = torch.as_tensor(x)
        outs = [self.encode_fused(xc) for xc in torch.chunk(x, bs_chunks, dim=0)]
        return [torch.cat(outs, dim=0)]

    def decode_layers(self, ids):
        """Layer-by-layer decode through the per-layer entry points (qpg_vq_gather_f32 + qpg_conv1d_f32);
        same result as decode(), kept to test those entry points."""
        ids = torch.as_tensor(ids).to(self.device, torch.int64).contiguous()
        B, L = ids.shape
        status = torch.zeros((1,), dtype=torch.int32, device=self.device)
        x = torch.empty((B, L, self.emb), dtype=torch.float32, device=self.device)
        _lib.call("qpg_vq_gather_f32", self.device, self.k, ids, B * L, self.emb, self.bins, x, status)
        T = L
        x = self._conv(self.dec_in, x, B, T, T, in_offset=-1)
        for res, even, odd in self.dec_up:
            x = self._resnet(res, x, B, T, self.reverse)
            y = torch.empty((B, 2 * T, even.cout), dtype=torch.float32, device=self.device)
            self._conv(even, x, B, T, T, in_offset=-1, out=y, out_stride=2, out_offset=0, T_y=2 * T)
            self._conv(odd, x, B, T, T, in_offset=0, out=y, out_stride=2, out_offset=1, T_y=2 * T)
            x, T = y, 2 * T
        return self._conv(self.dec_out, x, B, T, T, in_offset=-1)

    def decode(self, zs, start_level=0, end_level=None, bs_chunks=1):
        """VQVAE.decode (vqvae.py:152-159): zs = [LongTensor (B,L)] -> FloatTensor (B, 8L, C).
        The whole sequence is decoded in ONE convolutional pass like the reference
        (VisualizeCodebook.py:139-140): the dilated convolutions see across window seams."""
        assert self._loaded, "load_state_dict first"
        outs = []
        for ids in torch.chunk(torch.as_tensor(zs[0]), bs_chunks, dim=0):
            ids = ids.to(self.device, torch.int64).contiguous()
            B, L = ids.shape
            status = torch.zeros((1,), dtype=torch.int32, device=self.device)
            T = L * self.hop
            ws = self._workspace(B, T)
            out = torch.empty((B, T, self.input_dim), dtype=torch.float32, device=self.device)
            _lib.call("qpg_vq_decode_f32", self.device, self._desc, ids, B, L, ws, ws.numel(), out, status)
            outs.append(out)
            if int(status.item()):
                raise IndexError("code id out of range [0,%d)" % self.bins)
        return torch.cat(outs, dim=0)

    # convenience used by dataset_to_code / cal_distance equivalents -----------------------------
    def eval(self):
        return self

    @property
    def module(self):           # the reference calls model.module.encode(...) on a DataParallel wrapper
        return self


def normalize_poses(poses, data_mean, data_std):
    """(poses - mean) / clip(std, 0.01)  (make_beat_dataset.py:296-301)."""
    std = np.clip(np.asarray(data_std, np.float64).squeeze(), a_min=0.01, a_max=None)
    return (poses - np.asarray(data_mean, np.float64).squeeze()) / std


def dataset_to_code(model, poses, data_mean=None, data_std=None, batch=64):
    """process/make_beat_dataset.py::dataset_to_code (:261-325): (N,240,135) poses -> (N,30) int64 codes.
    The reference encodes one window at a time in a Python loop (:314-316); windows are independent, so
    they are encoded in batches here."""
    poses = np.asarray(poses)
    if data_mean is not None:
        poses = normalize_poses(poses.reshape(-1, poses.shape[-1]), data_mean, data_std).reshape(poses.shape)
    out = []
    for i in range(0, poses.shape[0], batch):
        x = torch.from_numpy(np.ascontiguousarray(poses[i:i + batch])).float()
        out.append(model.encode(x)[0].cpu().numpy())
    return np.concatenate(out, axis=0)


def cal_distance(model, n_codes=512, n_rep=30):
    """VisualizeCodebook.py::cal_distance (:93-116): decode [c]*30 for every code; signature = mean over the
    240 decoded frames (np.mean on the host, as the reference).  Returns dict(code, poses, signature)."""
    code = np.tile(np.arange(n_codes, dtype=np.int64)[:, None], (1, n_rep))
    poses = model.decode([torch.from_numpy(code)]).cpu().numpy()
    return dict(code=code, poses=poses, signature=np.mean(poses, axis=1))
