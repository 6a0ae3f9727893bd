"""Drop-in for step 3 of the reference's process/make_beat_dataset.py (:261-325, :604-607): encode every
240-frame pose window of a speaker database to 30 gesture codes.

    python -m qpgesture_amd.make_beat_dataset --config codebook.yml --save_dir ../dataset/BEAT \
        --prefix speaker_10_state_0 --gpu 0 --step 3 --VQVAE_model_path codebook_checkpoint_best.bin

Reads  <save_dir>/<prefix>/<prefix>_<split>_240.npz['body'] (N,240,135) for split in train/validation/test,
writes <save_dir>/<prefix>/<prefix>_<split>_240_code.npz['code'] int64 (N,30) — the files GestureKNN.py takes
as --train_codebook.  Same flags as codebook/configs/parse_args.py.  Steps 1, 2 and 4 (BVH/audio/text
preprocessing with third-party models) are out of scope.

The reference encodes one window per call in a Python loop (:314-316); windows are independent, so they are
encoded in batches of 256 on the GPU (qpg_vq_encode_f32)."""
import argparse
import os

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(description='Codebook')
    p.add_argument('--config', default='./configs/codebook.yml')
    p.add_argument('--gpu', type=str, default='0')
    p.add_argument('--no_cuda', type=list, default=['0'])
    p.add_argument('--prefix', type=str, required=False, default='speaker_10_state_0')
    p.add_argument('--save_path', type=str, required=False, default="./Speech2GestureMatching/output/")
    p.add_argument('--code_path', type=str, required=False)
    p.add_argument('--VQVAE_model_path', type=str, required=False)
    p.add_argument('--BEAT_path', type=str, default="../dataset/orig_BEAT/speakers/")
    p.add_argument('--save_dir', type=str, default="../dataset/BEAT")
    p.add_argument('--step', type=str, default="3")
    p.add_argument('--stage', type=str, default="train")
    p.add_argument('--batch', type=int, default=256)       # additive
    return p


def dataset_to_code(save_dir, prefix, n_frames=240, model_path=None, config=None, gpu="0", batch=256,
                    splits=("train", "validation", "test")):
    """dataset_to_code(save_dir, prefix, n_frames, model_path) of the reference (:261), same positional
    arguments; `config` = loaded codebook.yml (VQVAE hparams + data_mean/std)."""
    from .checkpoint import load_checkpoint, load_config
    from .vqvae import VQVAE, dataset_to_code as encode_windows
    if config is None:
        config = load_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "codebook.yml"))
    model = VQVAE(config.VQVAE, 15 * 9, device="cuda:%s" % gpu)
    model.load_state_dict(load_checkpoint(model_path)["model_dict"])
    written = {}
    for key in splits:
        src = os.path.join(save_dir, prefix, prefix + '_' + key + '_' + str(n_frames) + '.npz')
        if not os.path.exists(src):
            continue
        poses = np.load(src)['body']                                       # (N, 240, 135)   (:292)
        print(poses.shape)
        code = encode_windows(model, poses.reshape(-1, n_frames, poses.shape[-1]), config.data_mean,
                              config.data_std, batch=batch)                # normalise (:296-301) + encode (:316)
        dst = os.path.join(save_dir, prefix, prefix + '_' + key + '_' + str(n_frames) + '_code' + '.npz')
        np.savez_compressed(dst, code=code)                                # (:321-322)
        written[key] = dst
    return written


def main(argv=None):
    from .checkpoint import load_config
    args = build_parser().parse_args(argv)
    if args.step != "3":
        raise SystemExit("only --step 3 (dataset_to_code) is implemented; steps 1/2/4 are BVH/audio/text "
                         "preprocessing with third-party models, out of scope")
    return dataset_to_code(args.save_dir, args.prefix, 240, args.VQVAE_model_path, load_config(args.config),
                           gpu=args.gpu, batch=args.batch)


if __name__ == "__main__":
    main()
