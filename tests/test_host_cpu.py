"""CPU tests of the host logic and of the C-ABI library surface (no compute calls without a GPU)."""
import ctypes
import os
import tempfile

import numpy as np
import pytest

from qpgesture_amd import _lib, synth
from qpgesture_amd import code_knn as ck
from qpgesture_amd import data_processing as dp


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 16 and "qpg_audio_cosine_f64" in names and "qpg_match_steps" in names
    for n in names:
        assert hasattr(lib, n), "include/qpg.h declares %s but libqpg_hip.so does not export it" % n
    assert lib.qpg_version() >= 107
    import os
    assert lib.qpg_dev_kernarg() == (1 if os.environ.get("HIP_FORCE_DEV_KERNARG") == "1" else 0)


def test_bindings_cover_the_header():
    declared = set(_lib.declared_symbols())
    bound = set(_lib._SIGS) | {"qpg_version", "qpg_ctx_create", "qpg_ctx_destroy", "qpg_last_error",
                               "qpg_vq_workspace_floats", "qpg_vq_reduce_ws_bytes",
                               "qpg_conv1d_wgrad_ws_floats", "qpg_vq_code_sums_ws_bytes",
                               "qpg_text_percode_ws_bytes", "qpg_percode_select_mixed_ws_bytes",
                               "qpg_percode_select_mixed_ws_stride",
                               "qpg_merge_mixed_ws_bytes", "qpg_build_id", "qpg_ctx_set_option", "qpg_ctx_get_option",
                               "qpg_percode_select_exact_ws_bytes", "qpg_audio_hl_supported",
                               "qpg_audio_hl_db_bytes", "qpg_audio_hl_query_bytes", "qpg_hl_rows_bytes",
                               "qpg_hl_cols_bytes", "qpg_dev_kernarg", "qpg_audio_hl1_supported", "qpg_audio_hl1_db_bytes",
                               "qpg_conv16_image_bytes", "qpg_comm_unique_id", "qpg_comm_create", "qpg_comm_destroy"}
    assert declared == bound


def test_product_library_exports_no_debug_hooks_and_is_built_from_this_tree():
    """no global mutable state except the opaque context.  The measurement hooks (qpg_debug_*) exist only in
    -DQPG_DEBUG_HOOKS variant builds; the product library must not export any of them - and it must be the library compiled
    from THIS tree's sources: qpg_build_id() == the hash of csrc/* + include/qpg.h (build.source_hash)."""
    import subprocess
    from qpgesture_amd import build
    lib = _lib.load()
    hooks = _lib.debug_hook_symbols()
    assert "qpg_debug_gemm64_waves" in hooks and "qpg_debug_convt_shape" in hooks
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert "qpg_build_id" in exported and "qpg_audio_cosine_hl" in exported
    leaked = sorted(n for n in exported if n.startswith("qpg_debug"))
    assert not leaked, "the product library exports debug hooks: %s" % leaked
    for n in hooks:
        assert not hasattr(lib, n)
    assert lib.qpg_build_id().decode() == build.source_hash() == build.lib_build_id()
    # the knobs that remain are per-context (they need a device: exercised in the GPU suite); bad arguments fail cleanly here
    assert lib.qpg_ctx_set_option(None, 0, 1) == -1 and "null context" in _lib.last_error()


def test_error_reporting_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.qpg_ctx_create(0, ctypes.byref(h))
    assert rc < 0 and "qpg_ctx_create" in _lib.last_error()
    assert lib.qpg_ctx_create(0, None) == -1


def test_product_refuses_cpu_device():
    A = synth.make_db(2, 0, 64)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ck.GestureDB(synth.make_codes(2, 1), dp.interp_wavlm(A["wavlm"]), A["context"].squeeze(2),
                     A["phase_dense"], synth.make_signature(2), device="cpu")


def test_grids_match_oracle():
    from oracle import knn_oracle as O
    for n, st in ((180, 6), (398, 398 / 30), (240, 8)):
        ks, kint, cidx = O.audio_grid(n, st)
        a, b = ck.audio_grid(n, st)
        assert a == kint and b == cidx and len(a) == 26
    ks, rows = ck.text_grid()
    assert ks == list(range(0, 208, 8)) and rows == list(range(26))
    assert [ck.phase_slot(k) for k in (0, 6, 150, 200, 331)] == [O.phase_slot(k) for k in (0, 6, 150, 200, 331)]


def test_loader_matches_oracle_windowing():
    from oracle import knn_oracle as O
    with tempfile.TemporaryDirectory() as td:
        p = synth.write_npz_set(td, 3, 2, 7, 8, 9, 10, wavlm_dim=64)
        L = dp.load_db_codebook(p["train_database"], p["train_codebook"], p["test_data"], p["train_wavlm"],
                                p["test_wavlm"], p["train_wavvq"], p["test_wavvq"])
        tr = np.load(p["train_database"], allow_pickle=True)
        assert np.array_equal(L.train_wavlm, O.interp_wavlm(np.load(p["train_wavlm"])["wavlm"]))
        assert L.train_wavlm.shape == (3, 180, 64) and L.train_wavlm.flags.c_contiguous
        assert np.array_equal(L.train_phase, O.densify_phase(tr["phase"]))
        assert np.array_equal(L.train_phase, synth.make_db(3, 7, 64)["phase_dense"])
        assert L.train_context.shape == (3, 30, 384) and L.code.shape == (3, 30)
        assert L.test_wavvq.shape == (2, 398, 2)


def test_densify_phase_accepts_dense():
    x = np.random.default_rng(0).standard_normal((2, 240, 4, 8)).astype(np.float32)
    assert np.array_equal(dp.densify_phase(x), x)


def test_checkpoint_with_pickled_easydict(tmp_path):
    """train.py:114-116 pickles an easydict.EasyDict into the checkpoint; it must load without easydict."""
    import sys
    import types
    import torch
    from qpgesture_amd.checkpoint import AttrDict, load_checkpoint
    mod = types.ModuleType("easydict")
    EasyDict = type("EasyDict", (dict,), {"__module__": "easydict", "__qualname__": "EasyDict"})
    mod.EasyDict = EasyDict
    sys.modules["easydict"] = mod
    try:
        sd = {"module.bottleneck.level_blocks.0.k": torch.arange(6.).reshape(2, 3)}
        path = str(tmp_path / "ck.bin")
        torch.save({"args": EasyDict({"VQVAE": {"width": 512}, "lr": 3e-5}), "epoch": 7, "model_dict": sd}, path)
    finally:
        del sys.modules["easydict"]
    ck = load_checkpoint(path)
    assert ck["epoch"] == 7 and isinstance(ck["args"], AttrDict) and ck["args"].lr == 3e-5
    assert ck["args"].VQVAE.width == 512 and not hasattr(ck["args"], "dilation_cycle")
    assert torch.equal(ck["model_dict"]["module.bottleneck.level_blocks.0.k"], sd["module.bottleneck.level_blocks.0.k"])
    torch.save(sd, path)                                    # bare state_dict
    assert "model_dict" in load_checkpoint(path)


def test_reference_config_loads():
    import os
    from qpgesture_amd.checkpoint import load_config
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qpgesture_amd", "configs",
                     "codebook.yml")
    cfg = load_config(p)
    assert cfg.VQVAE.width == 512 and cfg.VQVAE.downs_t == [3] and len(cfg.data_mean) == 135


def test_training_host_logic_without_gpu():
    """Host-side pieces of the training loop that need no device: the reference-shaped initial state_dict
    (names/shapes of a fresh reference model, torch default init bounds), MultiStepLR, the CLI parser and config."""
    import torch
    from qpgesture_amd import synth, train
    from qpgesture_amd.checkpoint import load_config
    from qpgesture_amd.optim import MultiStepLR
    from qpgesture_amd.vqvae import init_state_dict
    sd = init_state_dict(seed=3)
    ref = {k[7:]: tuple(np.asarray(v).shape) for k, v in synth.make_vqvae_state_dict(7).items()}
    assert {k: tuple(v.shape) for k, v in sd.items()} == ref and len(sd) == 91
    w = sd["encoders.0.level_blocks.0.model.0.0.weight"]              # Conv1d(135, 512, 4): bound 1/sqrt(135*4)
    assert float(w.abs().max()) <= 1.0 / np.sqrt(135 * 4) and float(w.abs().max()) > 0.9 / np.sqrt(135 * 4)
    wt = sd["decoders.0.level_blocks.0.model.1.1.weight"]             # ConvTranspose1d(512, 512, 4): size(1)*k
    assert tuple(wt.shape) == (512, 512, 4) and float(wt.abs().max()) <= 1.0 / np.sqrt(512 * 4)
    assert float(sd["bottleneck.level_blocks.0.k"].abs().sum()) == 0.0
    assert torch.equal(init_state_dict(seed=3)["decoders.0.out.bias"], sd["decoders.0.out.bias"])

    class Opt:
        lr = 3e-5
    o = Opt()
    s = MultiStepLR(o, [100, 200], 0.1)
    lrs = []
    for _ in range(201):
        s.step()
        lrs.append(o.lr)
    assert lrs[98] == 3e-5 and abs(lrs[99] - 3e-6) < 1e-18 and abs(lrs[199] - 3e-7) < 1e-18

    a = train.parse_args(["--gpu", "1", "--synthetic", "8", "--epochs", "2"])
    assert a.gpu == "1" and a.synthetic == 8 and a.epochs == 2 and a.config.endswith("codebook.yml")
    cfg = load_config(a.config)
    assert cfg.batch_size == 256 and cfg.lr == 3e-5 and list(cfg.betas) == [0.5, 0.999] and cfg.VQVAE.l_mu == 0.99


def test_oracle_forward_loss_terms():
    """oracle/vqvae_oracle.losses against hand-computed values on a tiny sequence (vqvae.py:244-267)."""
    import torch
    from oracle import vqvae_oracle as VO
    xt = torch.tensor([[[0.0], [1.0], [3.0], [6.0]]])
    xo = torch.tensor([[[0.5], [1.0], [2.0], [6.0]]])
    loss, m = VO.losses(xt, xo, torch.tensor(2.0), hps_commit=0.02, vel=1.0, acc=1.0, reg=0.5)
    assert abs(float(m["recons_loss"]) - (0.5 + 0 + 1 + 0) / 4) < 1e-7
    # velocities: target [1,2,3], out [0.5,1,4] -> |diff| = [0.5,1,1]
    assert abs(float(m["velocity_loss"]) - 2.5 / 3) < 1e-7
    # accelerations: target [1,1], out [0.5,3] -> |diff| = [0.5,2]; regularisation = mean(out_acc^2)
    assert abs(float(m["acceleration_loss"]) - 1.25) < 1e-7
    assert abs(float(m["regularization"]) - (0.25 + 9.0) / 2) < 1e-6
    want = 0.375 + 2.0 * 0.02 + 0.5 * 4.625 + 2.5 / 3 + 1.25
    assert abs(float(loss) - want) < 1e-6


def test_ctypes_signatures_match_the_header():
    """Every binding in qpgesture_amd/_lib._SIGS has the argument count and the pointer / integer / float kinds of
    its prototype in include/qpg.h (a drifted ctypes signature would corrupt the call silently)."""
    import ctypes
    import re
    hdr = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|int64_t|void)\s+(qpg_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        protos[m.group(1)] = [re.sub(r"\s+", " ", a.strip()) for a in m.group(2).split(",")]

    def kind_of_c(arg):
        if "*" in arg:
            return "ptr"
        t = arg.split()[0] if " " in arg else arg
        if t in ("float",):
            return "f32"
        if t in ("double",):
            return "f64"
        if t in ("int64_t",):
            return "i64"
        return "i32"                                  # int, int32_t

    def kind_of_ct(t):
        if t is ctypes.c_void_p:
            return "ptr"
        return {ctypes.c_float: "f32", ctypes.c_double: "f64", ctypes.c_int64: "i64"}.get(t, "i32")
    for name, sig in _lib._SIGS.items():
        args = protos[name]
        assert [kind_of_c(a) for a in args[:2]] == ["ptr", "ptr"], name          # qpg_ctx*, void* stream
        want = [kind_of_c(a) for a in args[2:]]
        got = [kind_of_ct(t) for t in sig]
        assert got == want, (name, got, want)


# ---------------------------------------------------------------------------------------------------------------
# round 2 host logic
# ---------------------------------------------------------------------------------------------------------------
def test_tpack_layout_matches_its_definition():
    """vqvae.tpack: out[n // nb][kb][g][n % nb][j] = W[k = 16 kb + 4 g + j][n], k = tap * cin_pad + ci, zero padded
    (what csrc/qpg_convt.hip's fragment reads and LDS-DMA stages assume)."""
    import torch
    from qpgesture_amd.vqvae import tpack
    g = torch.Generator().manual_seed(0)
    taps, cin, cout, cin_pad, nb = 3, 20, 135, 32, 128
    w = torch.randn((taps, cin, cout), generator=g)
    p = tpack(w, cin_pad, nb).view(2, taps * cin_pad // 16, 4, nb, 4)
    for (n, tap, ci) in [(0, 0, 0), (134, 2, 19), (127, 1, 7), (128, 0, 16), (5, 2, 3)]:
        k = tap * cin_pad + ci
        assert p[n // nb, k // 16, (k % 16) // 4, n % nb, k % 4] == w[tap, ci, n]
    assert float(p[1, :, :, 7:, :].abs().max()) == 0.0                      # channels >= 135 are padding
    k = 0 * cin_pad + 25                                                     # input channel >= cin is padding
    assert float(p[:, k // 16, (k % 16) // 4, :, k % 4].abs().max()) == 0.0


def test_exchange_layout_offsets():
    from qpgesture_amd.code_knn import ExchangeLayout
    lay = ExchangeLayout(96, 512, 4, ["aud", "txt"], True, "cpu")
    n = 24 * 512
    # (round 4: every block ends with an 8-byte slot whose first i32 is the sender's trouble word)
    assert lay.Qb == 24 and lay.off == {"aud_d": 0, "aud_i": 8 * n, "txt_d": 12 * n, "txt_i": 16 * n, "flags": 20 * n}
    assert lay.block_bytes == 20 * n + 8 and lay.send.numel() == 4 * (20 * n + 8)
    d, i, qb, bs = lay.views("aud")
    assert d.dtype.is_floating_point and d.element_size() == 8 and d.numel() == n and (qb, bs) == (24, 20 * n + 8)
    one = ExchangeLayout(48, 512, 1, ["aud"], False, "cpu")                 # wavvq audio: f32 distances, all-gather
    assert one.views("aud")[0].element_size() == 4 and one.views("aud")[2:] == (0, 0)
    with pytest.raises(AssertionError):
        ExchangeLayout(50, 512, 4, ["aud"], True, "cpu")


def test_savgol_tables_and_bvh_writer(tmp_path):
    """bvh.savgol_tables == scipy.signal.savgol_filter(x, 15, 2) (mode='interp'), and the minimal BVH round-trips its
    channel table."""
    from scipy.signal import savgol_filter
    from qpgesture_amd import bvh
    mid, head, tail = bvh.savgol_tables(15, 2)
    x = np.random.default_rng(0).standard_normal(40)
    want = savgol_filter(x, 15, 2)
    got = np.array([head[t] @ x[:15] if t < 7 else (tail[t - 33] @ x[25:] if t >= 33 else mid @ x[t - 7:t + 8])
                    for t in range(40)])
    assert np.abs(got - want).max() < 1e-12
    e = np.random.default_rng(1).uniform(-170, 170, size=(5, 45))
    order = bvh.write_bvh(str(tmp_path / "x.bvh"), e)
    txt = open(str(tmp_path / "x.bvh")).read().splitlines()
    assert sorted(order) == list(range(15)) and order[0] == 0
    m = txt.index("MOTION")
    rows = np.array([l.split() for l in txt[m + 3:]], float)
    cols = np.concatenate([np.arange(3 * i, 3 * i + 3) for i in order])
    assert rows.shape == (5, 45) and np.abs(rows - e[:, cols]).max() < 1e-5
    assert txt[m + 1] == "Frames: 5" and txt.count("\tEnd Site") + sum("End Site" in l for l in txt) >= 3


def test_synth_variants_plant_what_they_say():
    from qpgesture_amd import synth
    tr, te, code = synth.make_db(48, 30), synth.make_db(2, 31), synth.make_codes(48, 32)
    synth.apply_variant(tr, te, code, "neartie")
    assert np.array_equal(te["wavlm"][0], tr["wavlm"][5]) and np.array_equal(tr["wavlm"][22], tr["wavlm"][5])
    d = tr["wavlm"][20] != tr["wavlm"][5]
    assert 0 < d.mean() < 0.02 and np.array_equal(code[20], code[5]) and not np.array_equal(code[21], code[5])
    assert np.abs(tr["wavlm"][20][d] / tr["wavlm"][5][d] - 1).max() < 2e-7          # one float32 ulp
    tr2, te2, code2 = synth.make_db(48, 40), synth.make_db(2, 41), synth.make_codes(48, 42)
    synth.apply_variant(tr2, te2, code2, "texttie")
    ctx = tr2["context"].reshape(-1, 384)
    assert len(np.unique(ctx, axis=0)) < 0.7 * len(ctx)                              # many identical rows
    with pytest.raises(ValueError):
        synth.apply_variant(tr, te, code, "nope")


def test_numpy_ranks_is_the_reference_expression():
    import torch
    from qpgesture_amd.code_knn import CodeKNN
    d = torch.tensor([[0.5, 0.0, 0.0, 1e3, 0.25], [3.0, 2.0, 1.0, 0.0, 1e3]], dtype=torch.float32)
    r = CodeKNN.numpy_ranks(d).numpy()
    for row, got in zip(d.numpy(), r):
        assert np.array_equal(got, np.array(list(row.astype(np.float64))).argsort().argsort())


def test_mixed_precision_constants_agree_with_the_header():
    """The band the mixed-precision select re-evaluates in must cover twice the sweep's bound as the C side states it."""
    import re
    from qpgesture_amd import code_knn
    txt = open(_lib.HEADER_PATH).read()
    m = re.search(r"#define\s+QPG_AUDIO_MX_ERR\s+([0-9.eE+-]+)", txt)
    assert m and float(m.group(1)) == code_knn.AUDIO_MX_ERR
    assert code_knn.AUDIO_MX_BAND >= 2.0 * code_knn.AUDIO_MX_ERR
    u = 2.0 ** -24
    assert code_knn.AUDIO_MX_ERR >= 32 * u / (1 - 32 * u) + 1e-13        # gamma_32 + the f64 part
    m2 = re.search(r"#define\s+QPG_AUDIO_HL_ERR\s+([0-9.eE+-]+)", txt)
    assert m2 and float(m2.group(1)) == code_knn.AUDIO_HL_ERR and code_knn.AUDIO_HL_BAND >= 2.0 * code_knn.AUDIO_HL_ERR
    # the split-f16 sweep's budget (csrc/qpg_audio_hl.hip, audio_cosine_hl2_kernel): chains of six instructions through
    # one accumulator, the four cross-term instructions first; representation; the f32-stored matrix;
    # subnormal l planes: an absolute error <= 2^-25 per element = 2^-25 sqrt(6144) / |x|_scaled of the product of norms per
    # side, with |x|_scaled >= sqrt(HL_NORM2_MIN) - the threshold the sweeps' validity guard enforces (the
    # guard used to ask for a scaled norm >= 1 while this line divided by 2^14); f64 sums
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qpgesture_amd", "csrc",
                            "qpg_audio_hl.hip")).read()
    norm2_min = float(re.search(r"#define\s+HL_NORM2_MIN\s+([0-9.eE+-]+)", src).group(1))
    assert len(re.findall(r"< HL_NORM2_MIN", src)) >= 2          # both operands of the sweep kernel
    k6 = 13 + 17 * 2.0 ** -10                       # cross instructions first: round 3's chain of two + a tiny C
    assert code_knn.AUDIO_HL_ERR >= (k6 * u + (2 * 2.0 ** -23 + u) + 2 * u +
                                     2 * 2.0 ** -25 * 6144 ** 0.5 / norm2_min ** 0.5 + 1e-13)
    # the one-plane image of an f16-stored track (round 5): the database side is exact - chains of four, the query's
    # representation and subnormals only
    assert code_knn.AUDIO_HL_ERR >= k6 * u + 2.0 ** -23 + 2 * u + 2.0 ** -25 * 6144 ** 0.5 / norm2_min ** 0.5 + 1e-13
    # the text prefilter / cfg-3 GEMM keeps round 3's kernel and bound (chains of two, separate cross accumulators)
    from qpgesture_amd import sorted_rows
    assert sorted_rows.HL_GEMM_ERR >= 13 * u + 12 * 193 * u / 2048 + (2 * 2.0 ** -23 + u) + 2 * u + 1e-13


def test_numpy_ranks_follow_the_reference_s_array_dtype():
    """the reference ranks np.array(list): float32 when every code has a (np.float32) text distance, float64 as
    soon as one `1e+3` Python float is left in the list, int64 for the Levenshtein audio (GestureKNN.py:553, 574)."""
    import torch
    from qpgesture_amd.code_knn import CodeKNN
    rng = np.random.default_rng(0)
    d = rng.standard_normal((3, 512)).astype(np.float32)
    d[:, 100:200] = d[:, :1]                                     # exact ties: the order NumPy's sort leaves is dtype business
    idx = np.zeros((3, 512), np.int32)
    idx[1, 7] = -1
    d[1, 7] = 1e3
    got = CodeKNN.numpy_ranks(torch.from_numpy(d), torch.from_numpy(idx)).numpy()
    for r in range(3):
        lst = [np.float32(x) for x in d[r]]
        if r == 1:
            lst[7] = 1e+3                                        # the reference's placeholder: a Python float
        arr = np.array(lst)
        assert arr.dtype == (np.float64 if r == 1 else np.float32)
        assert np.array_equal(got[r], arr.argsort().argsort())
    lev = rng.integers(0, 12, size=(2, 512)).astype(np.float32)
    got = CodeKNN.numpy_ranks(torch.from_numpy(lev), torch.zeros((2, 512), dtype=torch.int32), integer=True).numpy()
    for r in range(2):
        arr = np.array([int(x) for x in lev[r]])
        assert arr.dtype == np.int64 and np.array_equal(got[r], arr.argsort().argsort())


def test_status_word_is_never_ignored():
    from qpgesture_amd.code_knn import CodeKNN, GuardOverflow
    CodeKNN.check_status([0, 0])
    with pytest.raises(GuardOverflow) as e:
        CodeKNN.check_status([1, 5])                              # the guard's word outranks the IndexError
    assert e.value.flags == 5
    with pytest.raises(IndexError):
        CodeKNN.check_status([1, 0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset re-executes itself under torch.distributed.run with N ranks and
    still prints ONE JSON line (VERDICT r2 next #3a).  QPG_BENCH_LAUNCH_CHECK=1 stops every rank after the rendezvous
    (gloo all-reduce), before anything needs a GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QPG_BENCH_LAUNCH_CHECK="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"launch_check": True, "n_gpus": 3, "rank_sum": 6}


def test_audio_object_uses_m0_only_in_the_dma_asm():
    """qpg_audio.hip issues its LDS-DMA from inline asm that sets m0 (so that hipcc's wait-count model does not see the
    DMA) and names m0 as a clobber; that is safe as long as nothing the compiler generated reads or writes m0 in that
    translation unit.  Compile it to gfx950 assembly with the product's flags and check: every m0 operand belongs to an
    `s_mov_b32 m0` inside an ASMSTART / ASMEND block."""
    import os
    import subprocess
    import tempfile
    from qpgesture_amd import build as B
    src = os.path.join(B.CSRC, "qpg_audio.hip")
    if not os.path.exists(B.HIPCC):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "a.s")
        flags = [f for f in B.FLAGS if f not in ("-shared", "-Wall")]
        subprocess.check_call([B.HIPCC] + flags + ["-S", "--cuda-device-only", src, "-o", out], cwd=td,
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    inside, uses_out, movs = False, [], 0
    for l in lines:
        if "#ASMSTART" in l:
            inside = True
        elif "#ASMEND" in l:
            inside = False
        code = l.split(";")[0]
        if " m0" in code or ",m0" in code:
            if inside and "s_mov_b32 m0" in code:
                movs += 1
            else:
                uses_out.append(l)
    assert movs >= 30 and not uses_out, uses_out[:5]


def test_sorted_rows_drop_later_duplicates_of_a_code():
    """sorted_rows.SortedRows keeps only the FIRST of identical rows of a code (identical rows tie exactly for every query;
    first-wins picks the lowest index): the mask logic, on the CPU, with a forced hash collision path (same code, same
    hash is only possible for equal rows here, so collisions are emulated by equal hashes of unequal rows via a constant
    column layout)."""
    import torch
    from qpgesture_amd.sorted_rows import SortedRows
    torch.manual_seed(0)
    n, d = 300, 16
    x = torch.randn(n, d)
    codes = torch.randint(0, 5, (n,))
    x[10], codes[10] = x[3], codes[3]
    x[50], codes[50] = x[3], (codes[3] + 1) % 5             # same row, OTHER code: kept
    x[51], codes[51] = x[7], codes[7]
    x[120], codes[120] = x[119], codes[119]
    x[200], codes[200] = x[10], codes[10]                   # third copy
    keep = torch.tensor([i for i in range(n) if i != 119])  # 119 masked out: 120 is then the first of its kind
    m = SortedRows._first_of_duplicates(x, codes, keep)
    dropped = keep[~m].tolist()
    assert dropped == [10, 51, 200]
    # -0.0 and +0.0 are the same number: rows differing only there are duplicates
    y = torch.zeros(4, 8)
    y[1, 3] = -0.0
    y[2, 5] = 1.0
    m = SortedRows._first_of_duplicates(y, torch.zeros(4, dtype=torch.int64), torch.arange(4))
    assert m.tolist() in ([True, False, True, False], [True, True, True, False])   # (a -0.0 row may hash apart: kept)


def test_package_asks_for_device_kernargs_unless_the_user_chose():
    """qpgesture_amd/__init__.py: HIP_FORCE_DEV_KERNARG=1 by default (read by the HIP runtime at its initialisation), a
    value already in the environment is left alone."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os; %s; import qpgesture_amd; print(os.environ['HIP_FORCE_DEV_KERNARG'])"
    for prelude, want in (("os.environ.pop('HIP_FORCE_DEV_KERNARG', None)", "1"),
                          ("os.environ['HIP_FORCE_DEV_KERNARG'] = '0'", "0")):
        r = subprocess.run([sys.executable, "-c", code % prelude], cwd=root, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout.strip().splitlines()[-1] == want


def test_db_cache_file_format_roundtrip_on_the_host(tmp_path):
    """db_cache.py without a GPU: scalars, lists, nested dicts, NumPy arrays, host tensors, aliases and nested package
    objects survive save -> load bit for bit; a foreign key, a wrong magic and a short file are refused; the key follows
    the files' size / mtime and the options."""
    import torch
    from qpgesture_amd import db_cache
    from qpgesture_amd.code_knn import GestureDB
    from qpgesture_amd.sorted_rows import SortedRows
    db = object.__new__(GestureDB)
    sub = object.__new__(SortedRows)
    sub.__dict__.update(R=64, band=8.6e-5, row_code=torch.arange(64, dtype=torch.int16), device=torch.device("cpu"))
    t = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    db.__dict__.update(N=7, feature_dtype="f16", aud_k=[0, 6, 12], shape=(2, 3), hl_bound_report={"kappa": 8.3, "families": {"a": [1.0, 2.0]}},
                       freq_dist=np.linspace(0, 1, 5), code_host=np.arange(6, dtype=np.int64).reshape(2, 3), txt_r=t, txt_cidx=t,
                       txt_sorted=sub, hl_image=None, device=torch.device("cpu"), flag=True, lo=np.int64(3))
    p = str(tmp_path / "x.qpgdb")
    db_cache.save(db, p, "kk")
    assert db_cache.load(p, "cpu", "other") is None
    got = db_cache.load(p, "cpu", "kk")
    assert type(got) is GestureDB and type(got.txt_sorted) is SortedRows
    assert got.N == 7 and got.feature_dtype == "f16" and got.aud_k == [0, 6, 12] and got.shape == (2, 3) and got.flag is True
    assert got.lo == 3 and got.hl_image is None and got.hl_bound_report == db.hl_bound_report
    assert got.freq_dist.dtype == np.float64 and np.array_equal(got.freq_dist, db.freq_dist)
    assert np.array_equal(got.code_host, db.code_host) and torch.equal(got.txt_r, t) and got.txt_cidx is got.txt_r
    assert torch.equal(got.txt_sorted.row_code, sub.row_code) and got.txt_sorted.band == 8.6e-5
    assert got.device == torch.device("cpu") and got.txt_sorted.device == torch.device("cpu")
    raw = open(p, "rb").read()
    open(p, "wb").write(b"NOTQPGDB" + raw[8:])
    assert db_cache.load(p, "cpu", "kk") is None
    open(p, "wb").write(raw[:-100])
    assert db_cache.load(p, "cpu", "kk") is None
    f1 = tmp_path / "a.npz"
    f1.write_bytes(b"123")
    k1 = db_cache.file_key([str(f1)], {"tie_rule": "numpy"})
    assert k1 == db_cache.file_key([str(f1)], {"tie_rule": "numpy"}) != db_cache.file_key([str(f1)], {"tie_rule": "stable"})
    os.utime(str(f1), ns=(5, 5))
    assert db_cache.file_key([str(f1)], {"tie_rule": "numpy"}) != k1
    # round 6: a header that names a foreign module / a non-class, or that does not parse, is a MISS - never an
    # import of that module and never an exception on every later run
    import json
    open(p, "wb").write(raw)
    hl = int.from_bytes(raw[8:16], "little")
    head = json.loads(raw[16:16 + hl].decode())

    for bad in (dict(head, **{"class": "os:system"}), dict(head, **{"class": "qpgesture_amd.db_cache:VERSION"}),
                dict(head, **{"class": "nonsense"}), dict(head, tensors=[{"name": "x"}]), dict(head, attrs=7),
                dict(head, data_bytes="many")):
        hb = json.dumps(bad).encode()
        if len(hb) > hl:
            continue
        blob = raw[:8] + hl.to_bytes(8, "little") + hb + b" " * (hl - len(hb)) + raw[16 + hl:]   # (JSON ignores the padding)
        open(p, "wb").write(blob)
        assert db_cache.load(p, "cpu", "kk") is None, bad
    nested = json.loads(json.dumps(head))
    nested["attrs"]["txt_sorted"]["__object__"] = "subprocess:Popen"
    hb = json.dumps(nested).encode()
    open(p, "wb").write(raw[:8] + hl.to_bytes(8, "little") + hb + b" " * (hl - len(hb)) + raw[16 + hl:])
    assert db_cache.load(p, "cpu", "kk") is None
    open(p, "wb").write(raw[:16] + b"{" * hl + raw[16 + hl:])
    assert db_cache.load(p, "cpu", "kk") is None
    # save(): a failure leaves no tmp file behind; eviction by sources and by count
    db.bad = torch.arange(12).reshape(3, 4).t()                        # not contiguous
    with pytest.raises(ValueError):
        db_cache.save(db, p, "kk")
    assert not [n for n in os.listdir(str(tmp_path)) if ".tmp." in n]
    del db.__dict__["bad"]
    d2 = tmp_path / "cache"
    paths = [str(d2 / ("db_%02d.qpgdb" % i)) for i in range(4)]
    db_cache.save(db, paths[0], "k0", sources="S")
    db_cache.save(db, paths[1], "k1", sources="T")
    db_cache.save(db, paths[2], "k2", sources="S")                     # same sources as [0]: [0] goes
    assert sorted(os.listdir(str(d2))) == ["db_01.qpgdb", "db_02.qpgdb"]
    db_cache.save(db, paths[3], "k3", sources="U", keep=2)             # at most two files stay: the oldest ([1]) goes
    assert sorted(os.listdir(str(d2))) == ["db_02.qpgdb", "db_03.qpgdb"]
    assert db_cache.load(paths[3], "cpu", "k3") is not None


def test_c_abi_argument_checks_under_asan_ubsan():
    """the host side of the C ABI - argument checks, size computations, error formatting - built with
    AddressSanitizer + UndefinedBehaviorSanitizer (python -m qpgesture_amd.build --sanitize) and driven by
    tools/abi_sanitize.py: every entry point refuses null / zero / negative / huge arguments with an error code and a
    message, the size helpers survive the edges of their integer ranges, and neither sanitizer reports anything."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "abi_sanitize.py")], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "no sanitizer report" in r.stdout and "refused with a message" in r.stdout


def test_bench_sub_records_commands_parse():
    """bench.py's `sub_records` (round 6: BASELINE configs[2], [3] on one GPU, [4] in the driver's own line) are this script run
    again with other flags: every command must parse, must not recurse into further sub-records, and must name the
    configuration it claims."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    got = {}
    for name, what, flags in b.SUB_RUNS:
        a = b.build_parser().parse_args(flags + b.SUB_COMMON)
        assert a.no_sub_records and a.gpus == 1 and a.no_cpu_baseline and a.steps >= 20 and what
        got[name] = a
    assert got["cfg3"].workload == "cfg3"
    assert got["strong8192"].scaling == "strong" and got["strong8192"].n_db is None        # (strong: 8192 windows)
    c = got["clips16_f16_enc96"]
    assert (c.clips, c.feature_dtype, c.encode_batch, c.encode_precision) == (16, "f16", 96, "f32")
    assert got["clips16_f16_enc96_f16x3"].encode_precision == "f16x3"
    d = b.build_parser().parse_args([])                                                     # the driver's default run
    assert not d.no_sub_records and d.gpus == 1 and d.workload == "match"
