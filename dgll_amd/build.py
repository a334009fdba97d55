"""Build libdgll_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m dgll_amd.build [--force] [--verbose]

The shared library lands in dgll_amd/lib/ (git-ignored, but it travels with the gpurun snapshot).  hipcc
cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libdgll_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast", "-pthread",
         # `#pragma unroll` is a request, honoured up to this many instructions (default 16384): the transform kernels'
         # epilogues (4 tiles x 2 row groups, ragged paths included) need more, and a rolled loop puts their arrays in scratch
         "-mllvm", "-pragma-unroll-threshold=131072"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libdgll_hip.so cannot be built")
    return exe


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(HERE, "..", "include", "dgll_hip.h")]


STAMP = os.path.join(LIBDIR, "build_stamp.txt")


def source_digest():
    """sha256 over every kernel source and header (and the flags): identifies what the .so was built from."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in sources() + headers():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_current():
    try:
        with open(STAMP) as f:
            return os.path.exists(LIB) and f.read().strip() == source_digest()
    except OSError:
        return False


def build(force=False, verbose=False):
    """Compile + link under an exclusive file lock (torchrun imports the package on every rank at once: without it the
    ranks would run hipcc over the same objects and one could dlopen a half-linked library).  A rank that waited on the
    lock re-checks the stamp and returns without building."""
    import fcntl

    if not force and is_current():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    srcs, hdrs = sources(), headers()
    if not force and is_current():
        return LIB
    digest = source_digest()          # what is compiled now: an edit made WHILE hipcc runs must not be stamped as built
    if not srcs:
        raise RuntimeError("no HIP sources under " + CSRC)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = _newest(hdrs)

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time)):
            return obj, False
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, res.stdout, res.stderr))
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
        return obj, True

    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        tmp = LIB + ".tmp.%d" % os.getpid()
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", tmp] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
        os.replace(tmp, LIB)              # atomic: a concurrent dlopen sees the old or the new library, never a partial one
    with open(STAMP + ".tmp", "w") as f:
        f.write(digest)
    os.replace(STAMP + ".tmp", STAMP)     # the stamp goes last
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
