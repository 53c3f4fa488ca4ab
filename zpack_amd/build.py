"""Build every native piece of the repo for gfx950 (hipcc cross-compiles without a GPU).

  zpack_amd/libzpk_codec.so   HIP kernels + the codec C-ABI (include/zpack_codec.h)
  zpack_amd/libzpack_amd.so   the zpack.h host library (C) linked against the codec
  oracle/liboracle.so         CPU restatement (checker, tests only)
  oracle/_ref/libzpack_ref.so the compiled reference, when /root/reference is present
  benchdata/libzpkgen.so      synthetic archive generator (tests + bench input)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
CODEC_SO = os.path.join(HERE, "libzpk_codec.so")
ZPACK_SO = os.path.join(HERE, "libzpack_amd.so")
ARCH = "gfx950"


def _digest(sources, extra=""):
    """identity of a target's inputs: file names + contents (+ the flags); mtimes do not survive a push to a fresh box"""
    import hashlib
    h = hashlib.sha1(extra.encode())
    for s in sorted(sources):
        if s.endswith(".so"):                     # a linked-against library: its own stamp stands for it
            s = s + ".stamp" if os.path.exists(s + ".stamp") else s
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale(target, sources, extra=""):
    """True when the target is missing or was built from other inputs than the ones present (stamp file next to it)"""
    want = _digest(sources, extra)
    try:
        have = open(target + ".stamp").read().strip()
    except OSError:
        have = None
    return want if (have != want or not os.path.exists(target)) else None


def _stamp(target, digest):
    with open(target + ".stamp", "w") as fh:
        fh.write(digest + "\n")


def _hipcc():
    h = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(h):
        raise RuntimeError("hipcc not found: the MI355X codec cannot be built")
    return h


def build_codec(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "zpack_codec.h")]
    flags = (os.environ.get("ZPK_STATS") or "") + "|" + os.environ.get("ZPK_DEFINES", "")
    dig = _stale(CODEC_SO, srcs, flags)
    if not force and dig is None:
        return CODEC_SO
    dig = dig or _digest(srcs, flags)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-o", CODEC_SO, os.path.join(CSRC, "zpk_codec.hip")]
    if os.environ.get("ZPK_STATS"):          # developer build: per-phase cycle counters in the decode kernels
        cmd.insert(1, "-DZPK_STATS=1")
    for d in os.environ.get("ZPK_DEFINES", "").split():      # developer build: A/B variants (-D names)
        cmd.insert(1, "-D" + d)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    _stamp(CODEC_SO, dig)
    return CODEC_SO


def build_codec_dev(force=False, verbose=False):
    """zpack_amd/dev/libzpk_codec_dev.so: the same kernels with -DZPK_DEVELOPER (environment hooks: ZPK_WD_SCALE, ZPK_TRACE, ZPK_SKIP ...),
    loaded only by the tests that need a hook (ZPACK_AMD_CODEC_SO); the product library reads no environment in its launch path"""
    out = os.path.join(HERE, "dev", "libzpk_codec_dev.so")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "zpack_codec.h")]
    dig = _stale(out, srcs, "dev")
    if not force and dig is None:
        return out
    dig = dig or _digest(srcs, "dev")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-DZPK_DEVELOPER",
           "-o", out, os.path.join(CSRC, "zpk_codec.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    _stamp(out, dig)
    return out


def build_zpack(force=False, verbose=False):
    host = os.path.join(HERE, "host")
    csrcs = sorted(os.path.join(host, f) for f in os.listdir(host) if f.endswith(".c"))
    deps = csrcs + [os.path.join(ROOT, "include", "zpack.h"), os.path.join(ROOT, "include", "zpack_codec.h"), CODEC_SO]
    dig = _stale(ZPACK_SO, deps)
    if not force and dig is None:
        return ZPACK_SO
    dig = dig or _digest(deps)
    cmd = ["gcc", "-O2", "-g", "-fPIC", "-shared", "-std=c11", "-Wall", "-Wextra", "-D_FILE_OFFSET_BITS=64", "-D_POSIX_C_SOURCE=200809L", "-fvisibility=hidden",
           "-I" + os.path.join(ROOT, "include"), "-o", ZPACK_SO] + csrcs + \
          ["-L" + HERE, "-lzpk_codec", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    _stamp(ZPACK_SO, dig)
    return ZPACK_SO


def build_programs(force=False, verbose=False):
    """zpack_amd/zpk-batch: the batch-aware `t` / `x` commands over libzpack_amd.so (programs/zpk_batch.c)"""
    src = os.path.join(HERE, "programs", "zpk_batch.c")
    exe = os.path.join(HERE, "zpk-batch")
    deps = [src, ZPACK_SO, os.path.join(ROOT, "include", "zpack.h")]
    dig = _stale(exe, deps)
    if not force and dig is None:
        return exe
    dig = dig or _digest(deps)
    cmd = ["gcc", "-O2", "-g", "-std=c11", "-Wall", "-Wextra", "-D_FILE_OFFSET_BITS=64", "-I" + os.path.join(ROOT, "include"), "-o", exe, src,
           "-L" + HERE, "-lzpack_amd", "-lzpk_codec", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    _stamp(exe, dig)
    return exe


def build_helpers():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "benchdata"), "libzpkgen.so"])
    if os.path.isdir("/root/reference/lib"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
        if os.path.exists(CODEC_SO):         # INTEGRATION.md section B applied to a scratch copy of the reference, linked to the product codec
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "patched"])


def build_all(force=False, verbose=False):
    build_codec(force, verbose)
    build_codec_dev(force, verbose)
    if os.path.isdir(os.path.join(HERE, "host")):
        build_zpack(force, verbose)
        build_programs(force, verbose)
    build_helpers()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
