"""Reading the reference's VQ-VAE checkpoints and config.

train.py:114-116 saves `{'args': EasyDict, 'epoch': int, 'model_dict': DataParallel state_dict}` with
torch.save; the pickled `args` is an `easydict.EasyDict`, which need not be installed where the
checkpoint is read.  `load_checkpoint` unpickles it with a stand-in attribute-dict class.
"""
import pickle

import numpy as np
import torch
import yaml


class AttrDict(dict):
    """Minimal EasyDict: attribute access, AttributeError for missing keys (vqvae.py:73 uses hasattr)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __setstate__(self, state):
        for k, v in (state or {}).items():
            self[k] = v


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("easydict"):
            return AttrDict
        return super().find_class(module, name)


class _PickleShim:
    """pickle_module for torch.load: everything from pickle, our Unpickler."""
    __name__ = "qpg_pickle_shim"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


def load_checkpoint(path):
    """-> dict with 'model_dict' (and 'args', 'epoch' when present).  A bare state_dict is accepted too."""
    ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleShim)
    if isinstance(ck, dict) and "model_dict" in ck:
        return ck
    return {"model_dict": ck, "args": None, "epoch": None}


def load_config(path):
    """codebook/configs/codebook.yml -> AttrDict (VQVAE hparams, data_mean/std)."""
    import os
    with open(path) as f:
        cfg = AttrDict(yaml.safe_load(f))
    if "data_mean" not in cfg and "pose_stats" in cfg:            # (2,135) array [mean, std] next to the config
        stats = np.load(os.path.join(os.path.dirname(os.path.abspath(path)), cfg["pose_stats"]))
        cfg["data_mean"], cfg["data_std"] = [float(v) for v in stats[0]], [float(v) for v in stats[1]]
    return cfg


def denormalize_poses(poses, data_mean, data_std):
    """out*clip(std,0.01)+mean (VisualizeCodebook.py:124-126,148-149)."""
    std = np.clip(np.array(data_std).squeeze(), a_min=0.01, a_max=None)
    return np.multiply(poses, std) + np.array(data_mean).squeeze()
