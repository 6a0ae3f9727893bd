"""The matching + decoding half of codebook/Speech2GestureMatching/inference.py (:56-71): from the feature files of one
utterance to its gesture codes and decoded poses.

The reference's script first normalises / resamples the wav and extracts MFCC, energy, pitch and volume (librosa, sox,
ffmpeg-normalize: inference.py:19-54 - the audio front-end, out of this build's scope), then (:56-66)
runs `python GestureKNN.py` on the files it wrote and (:68-69) hands `knn_pred.npz` to VisualizeCodebook's
`visualizeCodeAndWrite`.  `main()` here starts where the front-end stops: it takes the `<name>_norm_mfcc.npz` the front-end
produced, issues the SAME GestureKNN command line (same flags; in-process by default, as a child process with
`subprocess=True` like the reference) and decodes the result with the VQ-VAE (`--stage inference` of VisualizeCodebook.py).
Nothing is computed on the host: both halves are the HIP paths of this package.

    python -m qpgesture_amd.inference --test_data X_norm_mfcc.npz --train_database D.npz --train_codebook C.npz \\
        --codebook_signature S.npz --train_wavlm WL.npz --test_wavlm TWL.npz --config codebook.yml --VQVAE_model_path CK.bin
"""
import argparse
import os
import subprocess as _subprocess
import sys


def knn_command(test_data, out_knn_filename, train_database, train_codebook, codebook_signature, train_wavlm, test_wavlm,
                out_video_path='./output/output_video_folder/', train_wavvq=None, test_wavvq=None, extra=()):
    """The argument list of inference.py:56-65, flag for flag (`--flag=value` form, as the reference spells it)."""
    cmd = ['--train_database=' + train_database,
           '--test_data=' + test_data,
           '--out_knn_filename=' + out_knn_filename,
           '--out_video_path=' + out_video_path,
           '--train_codebook=' + train_codebook,
           '--codebook_signature=' + codebook_signature,
           '--train_wavlm=' + train_wavlm,
           '--test_wavlm=' + test_wavlm]
    if train_wavvq:
        cmd.append('--train_wavvq=' + train_wavvq)
    if test_wavvq:
        cmd.append('--test_wavvq=' + test_wavvq)
    return cmd + list(extra)


def main(test_data, config, VQVAE_model_path, output_fold=None, prefix=None, gpu='0', subprocess=False, no_bvh=False,
         knn_extra=(), **db_paths):
    """test_data: the utterance's `<name>_norm_mfcc.npz` (inference.py:51-54 writes it).  db_paths: train_database,
    train_codebook, codebook_signature, train_wavlm, test_wavlm (, train_wavvq, test_wavvq) - inference.py:57-65.
    Writes `<output_fold>/knn_pred.npz` and `<output_fold>/result_<name>/generateresult_<name>.npy` (+ the BVH unless
    no_bvh), the reference's file names (:21-23, :68).  Returns (knn_pred int64 (M,30), poses f32 (240 M, 135))."""
    import numpy as np
    name = os.path.basename(test_data)
    for suffix in ('_norm_mfcc.npz', '_mfcc.npz', '.npz'):
        if name.endswith(suffix):
            name = name[:-len(suffix)]
            break
    output_fold = output_fold or os.path.dirname(os.path.abspath(test_data))
    os.makedirs(output_fold, exist_ok=True)
    out_knn = os.path.join(output_fold, 'knn_pred.npz')
    argv = knn_command(test_data, out_knn, extra=knn_extra, **db_paths)
    if subprocess:                                               # inference.py:66 subprocess.call(cmd)
        rc = _subprocess.call([sys.executable, '-m', 'qpgesture_amd.GestureKNN'] + argv)
        if rc != 0:
            raise RuntimeError('GestureKNN exited with %d' % rc)
    else:
        from . import GestureKNN
        GestureKNN.main(argv + ['--device', 'cuda:%s' % gpu])
    from . import VisualizeCodebook
    prefix = prefix or ('result_' + name)                        # inference.py:68
    vis = ['--config', config, '--gpu', str(gpu), '--code_path', out_knn, '--VQVAE_model_path', VQVAE_model_path,
           '--stage', 'inference', '--prefix', prefix, '--save_path', output_fold] + (['--no_bvh'] if no_bvh else [])
    poses, _ = VisualizeCodebook.main(vis)
    return np.load(out_knn)['knn_pred'], poses


def _cli(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for k in ('test_data', 'train_database', 'train_codebook', 'codebook_signature', 'train_wavlm', 'test_wavlm', 'config',
              'VQVAE_model_path'):
        p.add_argument('--' + k, required=True)
    p.add_argument('--train_wavvq')
    p.add_argument('--test_wavvq')
    p.add_argument('--output_fold')
    p.add_argument('--prefix')
    p.add_argument('--gpu', default='0')
    p.add_argument('--subprocess', action='store_true', help='run GestureKNN as a child process, as the reference does')
    p.add_argument('--no_bvh', action='store_true')
    a = p.parse_args(argv)
    kw = {k: v for k, v in vars(a).items() if v is not None}
    return main(**kw)


if __name__ == '__main__':
    _cli()
