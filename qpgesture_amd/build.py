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


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
