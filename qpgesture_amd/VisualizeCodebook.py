"""Drop-in for the decode-side stages of the reference's codebook/VisualizeCodebook.py (:373-393):

  --stage train      cal_distance (:93-116): decode every code x30 -> ./output/code.npz {code, poses, signature}
  --stage inference  visualize_code (:119-154): decode `knn_pred.flatten()` in ONE pass, de-normalise,
                     save <save_path>/<prefix>/generate<prefix>.npy and code<prefix>.npy

                     then (make_bvh_GENEA2020_BT's own arithmetic, process_bvh.py:57-76, on the device) the ZXY Euler
                     channel table <prefix>_euler.npy and a minimal <prefix>_generated.bvh carrying it
                     (--smoothing = its Savitzky-Golay option).

Same flags as codebook/configs/parse_args.py:4-18.  The pymo inverse pipeline (the recorded skeleton) and mp4
rendering stay out of scope (see qpgesture_amd/bvh.py).
Run: python -m qpgesture_amd.VisualizeCodebook --config ... --stage inference
"""
import argparse
import os

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(description='Codebook')
    p.add_argument('--config', default='./configs/codebook.yml')
    p.add_argument('--gpu', type=str, default='0')
    p.add_argument('--no_cuda', type=list, default=['0'])
    p.add_argument('--prefix', type=str, required=False, default='knn_pred_wavvq')
    p.add_argument('--save_path', type=str, required=False, default="./Speech2GestureMatching/output/")
    p.add_argument('--code_path', type=str, required=False)
    p.add_argument('--VQVAE_model_path', type=str, required=False)
    p.add_argument('--BEAT_path', type=str, default="../dataset/orig_BEAT/speakers/")
    p.add_argument('--save_dir', type=str, default="../dataset/BEAT")
    p.add_argument('--step', type=str, default="1")
    p.add_argument('--stage', type=str, default="train")
    p.add_argument('--signature_out', type=str, default='./output/code.npz')      # additive
    p.add_argument('--smoothing', action='store_true')                            # additive: process_bvh.py:62-68
    p.add_argument('--no_bvh', action='store_true')                               # additive: stop after the .npy files
    return p


def _model(cfg, model_path, gpu):
    from .checkpoint import load_checkpoint
    from .vqvae import VQVAE
    model = VQVAE(cfg.VQVAE, 15 * 9, device="cuda:%s" % gpu)
    model.load_state_dict(load_checkpoint(model_path)["model_dict"])
    return model


def main(argv=None):
    import torch
    from .checkpoint import denormalize_poses, load_config
    from .vqvae import cal_distance
    args = build_parser().parse_args(argv)
    cfg = load_config(args.config)
    model = _model(cfg, args.VQVAE_model_path, args.gpu)
    if args.stage == "train":
        out = cal_distance(model, n_codes=model.bins)
        os.makedirs(os.path.dirname(os.path.abspath(args.signature_out)), exist_ok=True)
        np.savez_compressed(args.signature_out, code=out["code"], poses=out["poses"], signature=out["signature"])
        return out
    if args.stage == "inference":
        code_source = np.load(args.code_path)['knn_pred']                            # :357
        zs = [torch.from_numpy(code_source.flatten()).unsqueeze(0)]                  # :139
        poses = model.decode(zs).squeeze(0).cpu().numpy()
        out_poses = denormalize_poses(poses, cfg.data_mean, cfg.data_std)           # :148-149
        out_code = np.vstack([zs[0].squeeze(0).numpy()])
        save_path = os.path.join(args.save_path, args.prefix)
        os.makedirs(save_path, exist_ok=True)
        np.save(os.path.join(save_path, 'code' + args.prefix + '.npy'), out_code)
        np.save(os.path.join(save_path, 'generate' + args.prefix + '.npy'), out_poses)
        print(out_poses.shape)
        print(out_code.shape)
        if args.no_bvh:
            return out_poses, out_code
        from . import bvh                                                          # :365 make_bvh_GENEA2020_BT
        euler = bvh.poses_to_euler(poses, cfg.data_mean, cfg.data_std, smoothing=args.smoothing,
                                   device="cuda:%s" % args.gpu)
        np.save(os.path.join(save_path, args.prefix + '_euler.npy'), euler)
        bvh.write_bvh(os.path.join(save_path, args.prefix + '_generated.bvh'), euler)
        return out_poses, out_code
    raise ValueError("stage must be train or inference")


if __name__ == "__main__":
    main()
