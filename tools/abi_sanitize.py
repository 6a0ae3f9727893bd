"""C-ABI argument checks under AddressSanitizer + UndefinedBehaviorSanitizer.

Parent mode (no arguments): builds qpgesture_amd/libqpg_hip_asan.so (python -m qpgesture_amd.build --sanitize) and re-runs
itself as a child with clang's ASan runtime preloaded and QPG_LIB_PATH pointing at the instrumented library.
Child mode: every entry point of include/qpg.h is called with arguments it must REFUSE - a null context, null pointers,
zero / negative / huge sizes - and has to come back with an error code and a message (no launch is ever made: the child
runs with the GPUs hidden - AMD's ASan runtime also intercepts the HSA allocator of a NON-instrumented HIP runtime and
aborts its first device allocation, which says nothing about this library), and the pure size helpers are driven to the
edges of their integer ranges.  Any report of either sanitizer aborts the child
(-fno-sanitize-recover, halt_on_error): the parent's exit code is the verdict."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from qpgesture_amd import _lib
    lib = _lib.load()
    assert "asan" in os.path.basename(_lib.LIB_PATH), _lib.LIB_PATH
    buf = ctypes.create_string_buffer(512)
    ctxs = [None]
    if torch.cuda.is_available():
        ctxs.append(_lib.ctx("cuda:0"))
    n_calls = n_refused = 0
    patterns = [lambda t: 0, lambda t: -1, lambda t: (1 << 30)]
    for ctx in ctxs:
        for name, sig in sorted(_lib._SIGS.items()):
            fn = getattr(lib, name)
            for pat in patterns:
                args = []
                for t in sig:
                    if t is ctypes.c_void_p:
                        args.append(None)
                    elif t in (ctypes.c_float, ctypes.c_double):
                        args.append(float(pat(t)))
                    elif t in (ctypes.c_int, ctypes.c_int32):
                        args.append(int(max(min(pat(t), 2 ** 31 - 1), -2 ** 31)))
                    else:
                        args.append(int(pat(t)))
                rc = fn(ctx, None, *args)
                n_calls += 1
                if rc != 0:
                    n_refused += 1
                    lib.qpg_last_error(buf, 512)
                    assert buf.value, name
                else:
                    # a call that SUCCEEDS on such arguments must be a documented no-op (empty problem: Q = 0 / N = 0)
                    assert ctx is not None and pat is patterns[0], "%s accepted invalid arguments" % name
    # size helpers at the edges of their ranges (signed overflow is UB: UBSan traps it)
    big = 2 ** 31 - 1
    for f, a in ((lib.qpg_audio_hl_db_bytes, (big, 1024)), (lib.qpg_audio_hl1_db_bytes, (big, 1024)),
                 (lib.qpg_audio_hl_query_bytes, (big, 1024)), (lib.qpg_hl_rows_bytes, (2 ** 40, 8192)),
                 (lib.qpg_hl_cols_bytes, (big, 8192)), (lib.qpg_percode_select_mixed_ws_bytes, (1 << 20, 512)),
                 (lib.qpg_percode_select_exact_ws_bytes, (1 << 16, 1 << 30, 512)), (lib.qpg_merge_mixed_ws_bytes, (1 << 20, 512, 1024)),
                 (lib.qpg_audio_hl_db_bytes, (0, 0)), (lib.qpg_audio_hl_db_bytes, (-5, -5)), (lib.qpg_hl_cols_bytes, (-1, 384))):
        f(*a)
        n_calls += 1
    for a in ((180, 1024, 26, 6, 2, 6), (0, 0, 0, 0, 0, 0), (-1, -1, -1, -1, -1, -1), (big, big, 26, 6, big, big)):
        lib.qpg_audio_hl_supported(*a)
        lib.qpg_audio_hl1_supported(*a)
    lib.qpg_last_error(buf, 1)
    lib.qpg_last_error(buf, 0)
    print("abi_sanitize: %d calls, %d refused with a message, contexts: %d (GPU: %s) - no sanitizer report"
          % (n_calls, n_refused, len(ctxs), torch.cuda.is_available()))


def main():
    if os.environ.get("QPG_ABI_SANITIZE_CHILD") == "1":
        return child()
    from qpgesture_amd import build
    lib = build.build_sanitized(verbose=False)
    env = dict(os.environ, QPG_ABI_SANITIZE_CHILD="1", QPG_LIB_PATH=lib, LD_PRELOAD=build.asan_runtime(),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-6000:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(main() or 0)
