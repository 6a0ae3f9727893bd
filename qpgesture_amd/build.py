"""Build libqpg_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libqpg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "qpg.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    for s in sources():
        o = s[:-4] + ".o"
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(
                os.path.getmtime(s), *(os.path.getmtime(os.path.join(CSRC, h)) for h in os.listdir(CSRC)
                                        if h.endswith(".h")),
                os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "qpg.h"))):
            cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


ASAN_LIB = os.path.join(HERE, "libqpg_hip_asan.so")


def asan_runtime():
    """Path of clang's shared ASan runtime (LD_PRELOAD for a non-instrumented host such as CPython)."""
    out = subprocess.check_output([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    if os.path.isabs(out) and os.path.exists(out):
        return out
    for root, _, files in os.walk("/opt/rocm/lib/llvm/lib/clang"):
        if "libclang_rt.asan-x86_64.so" in files:
            return os.path.join(root, "libclang_rt.asan-x86_64.so")
    raise RuntimeError("clang's ASan runtime not found")


def build_sanitized(verbose=True):
    """The HOST side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (the C-ABI's
    argument checks, size computations and error formatting are plain host C++); device code is compiled as usual
    (-fno-gpu-sanitize).  One hipcc call over all sources, -O1 -g: libqpg_hip_asan.so, used by tools/abi_sanitize.py and
    tests/test_host_cpu.py - never by the product."""
    if os.path.exists(ASAN_LIB) and not needs_build_for(ASAN_LIB):
        return ASAN_LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-Wno-unused-function",
           "-o", ASAN_LIB] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return ASAN_LIB


def needs_build_for(lib):
    t = os.path.getmtime(lib)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "qpg.h"))
    return any(os.path.getmtime(d) > t for d in deps)


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        build_sanitized()
    else:
        build_lib(force="--force" in sys.argv)
