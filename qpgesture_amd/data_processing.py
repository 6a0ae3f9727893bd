"""Database / query loading for the code-level matcher — the live part of the reference's
codebook/Speech2GestureMatching/data_processing.py::load_db_codebook (:197-353).

The reference expands every array into sliding-window feature stacks on the host (the WavLM one
is (N,180,6144) float64 = 8.85 MB per DB window).  Here only the BASE arrays are produced —
interpolated WavLM frames, squeezed context rows, dense phase — and the kernels address the
windows in place (qpg_audio_cosine_f64's tap offsets), so resident size is the base-array size.
The mfcc / energy / pitch / volume stacks the reference also builds are computed-but-unused in
the shipped mode and are not loaded.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .constant import num_frames_code


def interp_wavlm(wavlm, n_code=num_frames_code):
    """199 -> 180 frames per window: linear, align_corners=True, float32
    (data_processing.py:258-261).  Host-side torch call, identical to the reference's."""
    wavlm = np.ascontiguousarray(wavlm, np.float32)
    new_t = wavlm.shape[1] // n_code * n_code
    x = torch.from_numpy(wavlm).transpose(1, 2)
    y = F.interpolate(x, size=new_t, align_corners=True, mode="linear").transpose(1, 2)
    return np.ascontiguousarray(y.numpy())


def interp_wavlm_device(wavlm, device, n_code=num_frames_code, chunk=256):
    """Same resampling on the MI355X (qpg_wavlm_resample_f32, bit-exact with `interp_wavlm`): the raw
    (N,199,F) track is uploaded in chunks and only the (N,180,F) result stays resident.  Returns a device
    tensor, which GestureDB takes as-is (no host copy of the interpolated database)."""
    from . import _lib
    dev = torch.device(device)
    N, Tin, Fd = wavlm.shape
    Tout = Tin // n_code * n_code
    out = torch.empty((N, Tout, Fd), dtype=torch.float32, device=dev)
    for i in range(0, N, chunk):
        x = torch.from_numpy(np.ascontiguousarray(wavlm[i:i + chunk], np.float32)).to(dev)
        _lib.call("qpg_wavlm_resample_f32", dev, x, x.shape[0], Tin, Fd, Tout, out[i:i + chunk])
    return out


def densify_phase(phase):
    """The reference stores `phase` as an object array (n,240,4) of torch tensors shaped (1,8,1)
    (PAE.py:505-508, process/fix_device_bug.py:14-22).  Returns float32 (n,240,4,8); a dense
    float array of that shape (or (n,240,4,8,1,...)) is accepted too."""
    phase = np.asarray(phase)
    if phase.dtype != object:
        return np.ascontiguousarray(phase, np.float32).reshape(phase.shape[0], phase.shape[1], 4, 8)
    n, t, c = phase.shape
    out = np.empty((n, t, c, 8), np.float32)
    flat, o = phase.reshape(-1), out.reshape(-1, 8)
    for i in range(flat.shape[0]):
        v = flat[i]
        o[i] = (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)).reshape(8)
    return out


class LoadedDB(dict):
    __getattr__ = dict.__getitem__


def load_test_side(test_data_path, test_wavlm, test_wavvq, device=None):
    """The query side alone (the CLI with a prepared-database cache: db_cache.py): test interpolated WavLM, context,
    phase, wavvq - the arrays of load_db_codebook's `test_*` keys."""
    te = np.load(test_data_path, allow_pickle=True)
    out = LoadedDB()
    w = np.load(test_wavlm)["wavlm"]
    out["test_wavlm"] = interp_wavlm(w) if device is None else interp_wavlm_device(w, device)
    out["test_wavvq"] = np.load(test_wavvq)["wavvq"]
    out["test_context"] = np.ascontiguousarray(te["context"].squeeze(2), np.float32)
    return out


def load_db_codebook(data_file, codepath, test_data_path, train_wavlm, test_wavlm, train_wavvq, test_wavvq,
                     device=None):
    """Same arguments as the reference's load_db_codebook (+ `device`: resample the WavLM tracks on that GPU
    and return device tensors for them).  Returns the arrays the matcher needs:
    code (N,30); train/test interpolated WavLM (·,180,1024) f32; train/test context (·,30,384) f32;
    train/test phase (·,240,4,8) f32; train/test wavvq (·,398,2) ints."""
    tr = np.load(data_file, allow_pickle=True)
    te = np.load(test_data_path, allow_pickle=True)
    out = LoadedDB()
    out["code"] = np.load(codepath)["code"]
    if device is None:
        out["train_wavlm"] = interp_wavlm(np.load(train_wavlm)["wavlm"])
        out["test_wavlm"] = interp_wavlm(np.load(test_wavlm)["wavlm"])
    else:
        out["train_wavlm"] = interp_wavlm_device(np.load(train_wavlm)["wavlm"], device)
        out["test_wavlm"] = interp_wavlm_device(np.load(test_wavlm)["wavlm"], device)
    out["train_wavvq"] = np.load(train_wavvq)["wavvq"]
    out["test_wavvq"] = np.load(test_wavvq)["wavvq"]
    out["train_phase"] = densify_phase(tr["phase"])
    out["test_phase"] = densify_phase(te["phase"])
    out["train_context"] = np.ascontiguousarray(tr["context"].squeeze(2), np.float32)   # :342
    out["test_context"] = np.ascontiguousarray(te["context"].squeeze(2), np.float32)
    return out
