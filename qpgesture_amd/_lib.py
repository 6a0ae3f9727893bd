"""ctypes binding of libqpg_hip.so (the C ABI declared in include/qpg.h).

The library is the product: if it is missing or a call fails this module raises —
there is no CPU or PyTorch fallback anywhere in the package.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# QPG_LIB_PATH: kernel experiments only (a variant build of the same sources); the product is the in-tree library
LIB_PATH = os.environ.get("QPG_LIB_PATH") or os.path.join(_HERE, "libqpg_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "qpg.h")

_lib = None
_ctx = {}

c_void_p, c_int, c_int64, c_double, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_double, ctypes.c_float)
P, I, L = c_void_p, c_int, c_int64

# name -> argtypes (after ctx, stream)
_SIGS = {
    "qpg_signal_i32": [P, ctypes.c_int32],
    "qpg_doorbell_wait": [P, P, ctypes.c_int32],
    "qpg_wavlm_resample_f32": [P, L, I, I, I, P],
    "qpg_frame_norm2_f64": [P, L, I, P],
    "qpg_audio_cand_norm2": [P, I, I, P, I, I, I, P],
    "qpg_l2_normalize_rows_f32": [P, L, I, P],
    "qpg_text_pack_queries_f32": [P, I, I, I, P, P, I, P],
    "qpg_audio_pack_queries": [P, I, I, I, P, P, I, I, I, P, P],
    "qpg_audio_cosine_f64": [P, I, I, I, P, I, I, I, P, P, P, I, P, L],
    "qpg_audio_cosine_f64_h": [P, I, I, I, P, I, I, I, P, P, P, I, P, L],
    "qpg_audio_cosine_mx": [P, I, I, I, P, I, I, I, P, P, P, I, P, I, L, P],
    "qpg_audio_cosine_mx_h": [P, I, I, I, P, I, I, I, P, P, P, I, P, I, L, P],
    "qpg_audio_hl_pack_db": [P, I, I, I, I, I, I, I, P, L],
    "qpg_audio_hl_pack_queries": [P, I, I, P, L],
    "qpg_audio_pack_queries_hl": [P, I, I, I, P, P, I, I, I, P, P, P, L],
    "qpg_clip_pack_hl": [P, I, I, I, P, P, I, I, I, P, P, P, L, P, I, I, I, P, P, I, P, P, L],
    "qpg_audio_cosine_hl": [P, I, I, I, P, P, P, I, P, I, L, P],
    "qpg_audio_hl1_pack_db": [P, I, I, I, I, I, I, I, P, L],
    "qpg_audio_cosine_hl1": [P, I, I, I, P, P, P, I, P, I, L, P],
    "qpg_probe_mfma_f16_tile": [P, P, P, I, P],
    "qpg_hl_pack_rows": [P, L, I, P, L],
    "qpg_hl_pack_cols": [P, I, I, P, L],
    "qpg_hl_prepare_queries": [P, I, I, P, P, L, P],
    "qpg_hl_gemm_distance": [P, L, I, P, I, P, L, P, L],
    "qpg_percode_select_sorted_f32": [P, L, P, P, L, I, L, P, P, P, P, I, c_float, P, P, I, c_float, P, P, P, P, P,
                                      ctypes.c_int32, I, L],
    "qpg_hl_gemm_tilemin": [P, L, I, P, I, c_float, P, P, L],
    "qpg_hl_gemm_tilemin_h": [P, L, I, P, I, c_float, P, P, L],
    "qpg_perm32_rows_f32": [P, L, I, P],
    "qpg_percode_select_bycode_f32": [P, P, L, I, L, P, P, P, P, I, c_float, P, P, I, c_float, P, P, P, P, P, I],
    "qpg_text_pack_candidates_f32": [P, I, I, I, P, I, P],
    "qpg_text_cosine_f32": [P, L, I, P, I, P, L],
    "qpg_text_percode_f32": [P, L, I, P, I, P, I, I, ctypes.c_int32, c_float, P, L, P, P, P, P],
    "qpg_text_pack_candidates_f16": [P, I, I, I, P, I, P, P],
    "qpg_text_percode_f16": [P, P, L, I, P, I, P, I, ctypes.c_int32, c_float, P, L, P, P, P, P],
    "qpg_percode_select_f64": [P, L, I, P, L, I, c_double, ctypes.c_int32, P, P, P, I, L],
    "qpg_percode_select_f32": [P, L, I, P, L, I, c_float, ctypes.c_int32, P, P, P, I, L],
    "qpg_percode_select_guarded_f64": [P, L, I, P, L, I, c_double, ctypes.c_int32, P, P, P, I, L, P, I, I, P, I, I, I, P,
                                       c_double, P, I],
    "qpg_percode_select_mixed_f64": [P, I, L, I, P, L, I, c_double, ctypes.c_int32, P, P, P, I, L, P, I, I, P, I, I, I, P,
                                     P, P, c_double, c_double, P, P, L, I],
    "qpg_percode_select_mixed_f64_cut": [P, I, L, I, P, L, I, c_double, ctypes.c_int32, P, P, P, I, L, P, I, I, P, I, I, I,
                                         P, P, P, c_double, c_double, P, P, L, I, P, P, I, I],
    "qpg_percode_select_exact_f64": [P, L, I, P, L, I, c_double, ctypes.c_int32, P, P, P, I, L, P, I, I, P, I, I, I, P,
                                     c_double, P, I, P, L],
    "qpg_merge_mixed_phase1_f64": [P, I, L, L, L, I, I, c_double, c_double, I, P, L, P, L, P, I, L],
    "qpg_shard_refine_f64": [P, I, L, I, I, L, P, I, I, I, P, I, I, I, P, P, P, P, L, I, P, I],
    "qpg_flags_stamp": [P, I, L, L, P],
    "qpg_flags_gather": [P, I, L, L, P],
    "qpg_merge_mixed_phase2_f64": [P, I, L, L, I, I, c_double, P, L, P, L, P, P, P, P, I, c_double],
    "qpg_merge_select_f64": [P, I, L, L, L, I, I, c_double, P, P, P, c_double, P],
    "qpg_merge_select_f32": [P, I, L, L, L, I, I, c_float, P, P, P],
    "qpg_rank_rows_f64": [P, I, I, P],
    "qpg_rank_rows_f32": [P, I, I, P],
    "qpg_l2_table_f32": [P, I, I, P],
    "qpg_wavvq_lev_f32": [P, I, I, P, I, P, I, P, I, I, P, P, I, P, L],
    "qpg_conv1d_f32": [P, I, I, I, P, P, I, I, I, I, I, I, I, I, I, I, I, P, I, I, P, P, L],
    "qpg_convt_f32": [P, I, I, I, P, P, I, I, I, I, I, I, I, I, I, I, I, P, I, I, P],
    "qpg_convt_pair_f32": [P, I, I, I, P, P, I, I, P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "qpg_pad_channels_f32": [P, L, I, I, P],
    "qpg_conv16_pack_weights": [P, I, I, I, I, I, P, L],
    "qpg_conv16_f32": [P, I, I, I, I, P, I, P, I, I, I, I, I, I, I, I, I, P, I, I, P, P],
    "qpg_tpack_f32": [P, I, I, I, I, P],
    "qpg_resblock_f32": [P, I, I, I, P, P, P, P, P],
    "qpg_pose_to_euler_f64": [P, L, I, P, P, P, P, P, I, P, P],
    "qpg_vq_argmin_f32": [P, P, P, L, I, I, P, P, P],
    "qpg_vq_gather_f32": [P, P, L, I, I, P, P],
    "qpg_vq_encode_f32": [P, P, I, I, P, L, P, P, P],
    "qpg_vq_decode_f32": [P, P, I, I, P, L, P, P],
    "qpg_vq_loss_f32": [P, P, I, I, I, P, c_float, c_float, c_float, c_float, P, L, P],
    "qpg_vq_loss_grad_f32": [P, P, I, I, I, c_float, c_float, c_float, c_float, P],
    "qpg_vq_latent_stats_f32": [P, P, P, L, I, P, L, P],
    "qpg_vq_commit_grad_f32": [P, P, L, I, c_float, P, P],
    "qpg_vq_code_sums_f32": [P, P, L, I, I, P, P, P, L],
    "qpg_vq_ema_update_f32": [P, P, P, P, P, P, c_float, c_float, I, I, P, I, P, P, L, P],
    "qpg_conv1d_bwd_data_f32": [P, I, I, I, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P, P, P, P, L],
    "qpg_conv1d_bwd_weight_f32": [P, I, I, I, P, I, I, I, I, I, I, I, I, I, I, I, I, P, P, I, P, L],
    "qpg_adam_step_f32": [P, P, P, P, L, c_float, c_float, c_float, c_float, L],
    "qpg_comm_allgather": [P, P, P, L],
    "qpg_comm_alltoall": [P, P, P, L],
    "qpg_comm_allreduce_max_i32": [P, P, L],
    "qpg_allreduce_min_u64": [P, P, L],
    "qpg_pack_min_u64": [P, P, L, P],
    "qpg_unpack_min_u64": [P, L, c_float, P, P],
    "qpg_match_steps": [P, P, P, P, P, P, P, I, P, P, I, P, P, I, P, I, I, I, I, I, I, P, P, P, P, P, P, P],
    "qpg_match_steps_batch": [P, P, P, P, P, P, P, I, P, P, I, P, P, I, P, I, I, I, I, I, I, P, P, P, P, P, P, P, L, P],
    "qpg_fuse_best_ranked": [P, P, P, P, I, I, P],
}


QPG_VQ_MAX_DOWN, QPG_VQ_MAX_DEPTH = 4, 4


class ConvDesc(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("taps", ctypes.c_int32), ("cin", ctypes.c_int32),
                ("cin_pad", ctypes.c_int32), ("cout", ctypes.c_int32), ("cout_pad", ctypes.c_int32),
                ("wt", c_void_p)]


class VqModel(ctypes.Structure):
    """qpg_vq_model of include/qpg.h."""
    _fields_ = [("in_dim", ctypes.c_int32), ("width", ctypes.c_int32), ("emb", ctypes.c_int32),
                ("bins", ctypes.c_int32), ("down_t", ctypes.c_int32), ("depth", ctypes.c_int32),
                ("growth", ctypes.c_int32), ("reverse_dec", ctypes.c_int32),
                ("enc_down", ConvDesc * QPG_VQ_MAX_DOWN),
                ("enc_res", ((ConvDesc * 2) * QPG_VQ_MAX_DEPTH) * QPG_VQ_MAX_DOWN),
                ("enc_out", ConvDesc), ("dec_in", ConvDesc),
                ("dec_res", ((ConvDesc * 2) * QPG_VQ_MAX_DEPTH) * QPG_VQ_MAX_DOWN),
                ("dec_up_even", ConvDesc * QPG_VQ_MAX_DOWN), ("dec_up_odd", ConvDesc * QPG_VQ_MAX_DOWN),
                ("dec_out", ConvDesc), ("kT", ConvDesc), ("k", c_void_p), ("kk", c_void_p),
                ("enc_res_pack", (c_void_p * QPG_VQ_MAX_DEPTH) * QPG_VQ_MAX_DOWN),
                ("dec_res_pack", (c_void_p * QPG_VQ_MAX_DEPTH) * QPG_VQ_MAX_DOWN)]


_HOOKS_RE = re.compile(r"#ifdef QPG_DEBUG_HOOKS\n(.*?)#endif", re.S)


def declared_symbols():
    """Every function name include/qpg.h declares for the PRODUCT library (used by the CPU symbol-export test); the
    #ifdef QPG_DEBUG_HOOKS block - experiment builds only - is debug_hook_symbols()."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = _HOOKS_RE.sub("", txt)
    return sorted(set(re.findall(r"\b(qpg_[a-z0-9_]+)\s*\(", txt)))


def debug_hook_symbols():
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(n for blk in _HOOKS_RE.findall(txt) for n in re.findall(r"\b(qpg_[a-z0-9_]+)\s*\(", blk)))


def load():
    """dlopen the library (works without a GPU: symbols only)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libqpg_hip.so is missing (%s). Build it with `python -m qpgesture_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
    if not os.environ.get("QPG_LIB_PATH"):
        # the in-tree library must be the one compiled from THIS tree's sources (a stale prebuilt would be tested and
        # benchmarked as if it were HEAD): compare its stamped id with the tree's hash BEFORE mapping it; rebuild on a mismatch
        from . import build as _build
        want = _build.source_hash()
        if _build.lib_build_id(LIB_PATH) != want:
            _build.build_lib(verbose=False)
    lib = ctypes.CDLL(LIB_PATH)
    lib.qpg_version.restype = c_int
    lib.qpg_build_id.restype = ctypes.c_char_p
    lib.qpg_build_id.argtypes = []
    lib.qpg_ctx_set_option.argtypes = [c_void_p, c_int, c_int]
    lib.qpg_ctx_get_option.argtypes = [c_void_p, c_int, ctypes.POINTER(c_int)]
    lib.qpg_ctx_create.argtypes = [c_int, ctypes.POINTER(c_void_p)]
    lib.qpg_ctx_create.restype = c_int
    lib.qpg_ctx_destroy.argtypes = [c_void_p]
    lib.qpg_last_error.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.qpg_vq_workspace_floats.argtypes = [c_void_p, c_int, c_int]
    lib.qpg_vq_workspace_floats.restype = c_int64
    lib.qpg_conv1d_wgrad_ws_floats.argtypes = [c_int, c_int, c_int, c_int]
    lib.qpg_conv1d_wgrad_ws_floats.restype = c_int64
    lib.qpg_vq_code_sums_ws_bytes.argtypes = [c_int64, c_int, c_int]
    lib.qpg_vq_code_sums_ws_bytes.restype = c_int64
    lib.qpg_text_percode_ws_bytes.argtypes = [c_int64, c_int, c_int, c_int]
    lib.qpg_text_percode_ws_bytes.restype = c_int64
    lib.qpg_percode_select_mixed_ws_bytes.argtypes = [c_int, c_int]
    lib.qpg_percode_select_mixed_ws_bytes.restype = c_int64
    lib.qpg_percode_select_mixed_ws_stride.argtypes = [c_int]
    lib.qpg_percode_select_mixed_ws_stride.restype = c_int64
    lib.qpg_merge_mixed_ws_bytes.argtypes = [c_int, c_int, c_int]
    lib.qpg_audio_hl_supported.argtypes = [c_int] * 6
    lib.qpg_audio_hl1_supported.argtypes = [c_int] * 6
    lib.qpg_audio_hl1_db_bytes.argtypes = [c_int, c_int]
    lib.qpg_audio_hl1_db_bytes.restype = c_int64
    lib.qpg_audio_hl_db_bytes.argtypes = [c_int, c_int]
    lib.qpg_audio_hl_db_bytes.restype = c_int64
    lib.qpg_audio_hl_query_bytes.argtypes = [c_int, c_int]
    lib.qpg_audio_hl_query_bytes.restype = c_int64
    lib.qpg_hl_rows_bytes.argtypes = [c_int64, c_int]
    lib.qpg_hl_rows_bytes.restype = c_int64
    lib.qpg_hl_cols_bytes.argtypes = [c_int, c_int]
    lib.qpg_hl_cols_bytes.restype = c_int64
    lib.qpg_percode_select_exact_ws_bytes.argtypes = [c_int, c_int64, c_int]
    lib.qpg_percode_select_exact_ws_bytes.restype = c_int64
    lib.qpg_merge_mixed_ws_bytes.restype = c_int64
    lib.qpg_conv16_image_bytes.argtypes = [c_int, c_int, c_int]
    lib.qpg_conv16_image_bytes.restype = c_int64
    lib.qpg_comm_unique_id.argtypes = [ctypes.c_char_p, c_int64]
    lib.qpg_comm_create.argtypes = [c_void_p, ctypes.c_char_p, c_int64, c_int, c_int, ctypes.POINTER(c_void_p)]
    lib.qpg_comm_destroy.argtypes = [c_void_p]
    for name, at in (("qpg_debug_convt_shape", [c_int, c_int]), ("qpg_debug_convt_opts", [c_int, c_int]),
                     ("qpg_debug_gemm64_waves", [c_int])):      # -DQPG_DEBUG_HOOKS variant libraries only (QPG_LIB_PATH)
        if hasattr(lib, name):
            getattr(lib, name).argtypes = at
    lib.qpg_vq_reduce_ws_bytes.argtypes = []
    lib.qpg_vq_reduce_ws_bytes.restype = c_int64
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = [c_void_p, c_void_p] + sig
        fn.restype = c_int
    _lib = lib
    return lib


QPG_OPT_GATE_DEDUP_FROM_CHAINS = 0


def set_option(device, option, value):
    """qpg_ctx_set_option on `device`'s context (per-context knob, include/qpg.h)."""
    rc = load().qpg_ctx_set_option(ctx(device), option, value)
    if rc != 0:
        raise RuntimeError("qpg_ctx_set_option failed (%d): %s" % (rc, last_error()))


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().qpg_last_error(buf, 512)
    return buf.value.decode()


def ctx(device):
    """Per-device opaque context (created on first use)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _ctx:
        h = c_void_p()
        rc = load().qpg_ctx_create(idx, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("qpg_ctx_create(%d) failed: %s" % (idx, last_error()))
        _ctx[idx] = h
    return _ctx[idx]


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return c_void_p(t.data_ptr())


_cur_dev = None
_dev_index = {}
n_calls = 0          # launches made through call() / prepare() (parallel.SegmentRecorder: is a captured segment empty?)
# torch's C entry points behind torch.cuda.current_device() / current_stream().cuda_stream: the Python wrappers cost ~1 and
# ~3 us per call (lazy-init checks, a Stream object per call) and this module makes a dozen calls per 0.36 ms step
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_stream(idx):
    if _raw is not None:
        return _raw(idx)
    return torch.cuda.current_stream(idx).cuda_stream



def call(name, device, *args):
    """Invoke a C-ABI entry point on torch's current stream of `device`; raise on error.

    HIP launches go to the CURRENT device (a stream handle of 0 means "the current device's default stream"),
    so when `device` is not the calling thread's current device the call is made under torch.cuda.device(device):
    a GestureDB / VQVAE built on cuda:1 works whatever device the caller has selected.
    (This wrapper sits in front of every launch of a clip - a dozen per 0.4 ms step - so it avoids what it can:
    device indices are cached, tensors go in as their data_ptr() integers.)"""
    global n_calls
    lib = _lib if _lib is not None else load()
    idx = _dev_index.get(device)
    if idx is None:
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        _dev_index[device] = idx
    if idx != _get_device():
        with torch.cuda.device(idx):
            return call(name, device, *args)
    n_calls += 1
    stream = _raw_stream(idx)
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            if not (a.is_cuda and a.is_contiguous()):
                raise AssertionError("device-resident contiguous tensor required")
            conv.append(a.data_ptr())
        elif isinstance(a, ctypes.Structure):
            conv.append(ctypes.byref(a))
        else:
            conv.append(a)                      # (ints, floats, None, c_void_p handles)
    h = _ctx.get(idx)
    if h is None:
        h = ctx(device)
    rc = getattr(lib, name)(h, stream, *conv)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, last_error()))


def prepare(name, device, *args):
    """call() in two halves: everything but the foreign call happens now, the returned function makes the launch (on the
    stream that is current NOW).  For a launch that has to follow another one closely - the sweep behind its query pack:
    the argument conversion of its 15 arguments would otherwise sit between the two launches (~5 us of idle GPU)."""
    lib = _lib if _lib is not None else load()
    idx = _dev_index.get(device)
    if idx is None or idx != _get_device():
        return lambda: call(name, device, *args)
    stream = _raw_stream(idx)
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            if not (a.is_cuda and a.is_contiguous()):
                raise AssertionError("device-resident contiguous tensor required")
            conv.append(a.data_ptr())
        elif isinstance(a, ctypes.Structure):
            conv.append(ctypes.byref(a))
        else:
            conv.append(a)
    h = _ctx.get(idx)
    if h is None:
        h = ctx(device)
    fn = getattr(lib, name)

    def launch(_keep=args):          # (the tensors stay alive until the launch has been made)
        global n_calls
        n_calls += 1
        rc = fn(h, stream, *conv)
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (name, rc, last_error()))
    return launch
