"""Grid constants of the code-level matching path.

Mirrors the values (not the text) of the reference's
codebook/Speech2GestureMatching/constant.py:14-40; only the ones the
CodeKNN path reads are kept.
"""

NUM_AUDIO_FEAT_FRAMES = 6   # taps per audio feature window      (constant.py:14)
NUM_MFCC_FEAT = 13          #                                     (constant.py:20)
STEP_SZ = 4                 # codes appended per matching step    (constant.py:24)
FRAME_INTERVAL = 4          # mfcc tap stride; wavlm uses -2 = 2  (constant.py:26)
num_frames = 240            # pose frames per DB window           (constant.py:38)
num_frames_code = 30        # codes per DB window                 (constant.py:39)
codebook_size = 512         #                                     (constant.py:40)

WAVVQ_FRAMES = 398          # vq-wav2vec frames per 4 s window (GestureKNN.py:436-438)
WAVVQ_GROUPS = 2
WAVVQ_GROUP_SIZE = 320      # symbol = g1*320+g2                  (GestureKNN.py:60)
PHASE_CHANNELS = 8          # PAE latent phase channels           (GestureKNN.py:451)
ABSENT_DIST = 1e+3          # initial per-code distance           (GestureKNN.py:668,709)
SEED = 123456               # GestureKNN.py:19-22
