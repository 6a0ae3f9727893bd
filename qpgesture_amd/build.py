"""Build libqpg_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
import hashlib
import json
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


MANIFEST = os.path.join(CSRC, ".build_manifest.json")     # object file -> hash of what it was compiled from (untracked)
ID_SOURCE = "qpg_core.hip"                                  # the one file that is compiled with -DQPG_BUILD_ID


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(os.path.dirname(HERE), "include", "qpg.h")]


def _digest(paths, extra=b""):
    h = hashlib.sha256(extra)
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def source_hash():
    """What qpg_build_id() of a library built from THIS tree returns: SHA-256 over csrc/*.hip, csrc/*.h and include/qpg.h
    (contents, in name order), first 16 hex digits.  Content, not mtime: a checkout, a copy to the GPU box or a `touch`
    changes nothing, an edit does."""
    return _digest(sources() + headers())


def lib_build_id(lib=LIB):
    """qpg_build_id() of an existing library file, None if it has none (missing file, a pre-round-6 build).  Read from the
    file's bytes (the "QPG_BUILD_ID=<id>" marker qpg_core.hip embeds), not through dlopen: a library that this process has
    already mapped would answer for the OLD file after a rebuild."""
    try:
        with open(lib, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(b"QPG_BUILD_ID=")
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + 13:j].decode(errors="replace")


def needs_build():
    return lib_build_id() != source_hash()


def _object_hash(src, build_id):
    flags = " ".join(FLAGS).encode()
    if os.path.basename(src) == ID_SOURCE:
        flags += b" " + build_id.encode()
    return _digest([src] + headers(), flags)


def build_lib(force=False, verbose=True):
    """Compile what changed (by CONTENT: csrc/.build_manifest.json holds, per object, the hash of its source + the headers
    + the flags it was compiled with) and link; qpg_core.hip carries the tree's source_hash() as QPG_BUILD_ID, so the
    library can say which sources it was built from."""
    bid = source_hash()
    if not force and lib_build_id() == bid:
        return LIB
    try:
        with open(MANIFEST) as f:
            manifest = json.load(f)
    except (OSError, ValueError):
        manifest = {}
    objs = []
    for s in sources():
        o = s[:-4] + ".o"
        want = _object_hash(s, bid)
        if force or not os.path.exists(o) or manifest.get(os.path.basename(o)) != want:
            cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", s, "-o", o]
            if os.path.basename(s) == ID_SOURCE:
                cmd.insert(1, '-DQPG_BUILD_ID="%s"' % bid)
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            manifest[os.path.basename(o)] = want
            with open(MANIFEST, "w") as f:
                json.dump(manifest, f, indent=0, sort_keys=True)
        objs.append(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    got = lib_build_id()
    if got != bid:
        raise RuntimeError("libqpg_hip.so reports build id %r, the tree hashes to %r" % (got, bid))
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
