"""Build libmpnn_amd.so (the C-ABI library) in-tree with hipcc for gfx950.

    python -m mpnn_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the source snapshot.

The library is STALE when the manifest written next to it at link time (a hash over every source, every header and the
flag list) differs from the hash of the files now in the tree -- content, not mtimes: a snapshot copied to the GPU box
keeps its library, an edited .hip does not.  _lib.load() checks this on every first load and rebuilds, or raises when
hipcc is absent.  An object is recompiled when its source or any header is newer, or when it was compiled with other
flags.  Objects and the library are written to a per-process temporary name and renamed into place, and the whole build
holds a file lock, so several ranks that find the library stale at start-up build it once instead of over each other.

Builds with extra flags (MPNN_EXTRA_HIPCC_FLAGS: timing experiments, some with wrong results by design) go to a
directory of their own, lib/variant_<hash of the flags>/, and never replace the product library.
"""
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ARCH = "gfx950"
EXTRA = os.environ.get("MPNN_EXTRA_HIPCC_FLAGS", "").split()
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-fno-fast-math"] + EXTRA
FLAGS_HASH = hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:16]
LIBDIR = os.path.join(HERE, "lib", "variant_" + FLAGS_HASH) if EXTRA else os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmpnn_amd.so")
MANIFEST = LIB + ".manifest"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "mpnn_amd.h")]


def _obj_of(src):
    return os.path.join(LIBDIR, "obj", os.path.basename(src)[:-4] + ".o")


def _obj_stale(src, t_hdr):
    obj = _obj_of(src)
    if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_hdr):
        return True
    try:
        with open(obj + ".flags") as f:
            return f.read().strip() != FLAGS_HASH
    except OSError:
        return True


def source_hash():
    """Hash over the flag list and the contents of every source and header: what a library must have been built from."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(sources() + _headers(), key=os.path.basename):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _stale():
    """True when the library is missing or was not built from the sources, headers and flags now in the tree."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(MANIFEST) as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def file_flags(src):
    """Extra hipcc flags a source asks for in a `// hipcc-flags: ...` line among its first lines (they are part of the file,
    so the content hash of the sources covers them)."""
    with open(src) as f:
        head = [next(f, "") for _ in range(8)]
    out = []
    for line in head:
        if line.startswith("// hipcc-flags:"):
            out += line[len("// hipcc-flags:"):].split("(")[0].split()
    return out


def _compile_one(job):
    hipcc, src, obj, verbose = job
    tmp = "%s.%d.tmp" % (obj, os.getpid())
    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + file_flags(src) + ["-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode == 0:
        os.replace(tmp, obj)
        with open(obj + ".flags.tmp%d" % os.getpid(), "w") as f:
            f.write(FLAGS_HASH + "\n")
        os.replace(obj + ".flags.tmp%d" % os.getpid(), obj + ".flags")
    elif os.path.exists(tmp):
        os.remove(tmp)
    return src, r.returncode, r.stdout + r.stderr


def build(force=False, verbose=False):
    """Compile every HIP source to an object (only the stale ones, a few in parallel) and link them into one shared
    object; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmpnn_amd.so (and there is no CPU fallback)")
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():                    # another process built it while this one waited
            return LIB
        t_hdr = max(os.path.getmtime(p) for p in _headers())
        jobs, objs = [], []
        for src in sources():
            objs.append(_obj_of(src))
            if force or _obj_stale(src, t_hdr):
                jobs.append((hipcc, src, _obj_of(src), verbose))
        if jobs:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("MPNN_BUILD_JOBS", "6")))) as ex:
                for src, rc, log in ex.map(_compile_one, jobs):
                    if rc != 0:
                        sys.stderr.write(log)
                        raise RuntimeError("hipcc failed on %s" % src)
                    if verbose and log:
                        sys.stderr.write(log)
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp] + objs, capture_output=True,
                           text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed linking libmpnn_amd.so")
        os.replace(tmp, LIB)
        with open(MANIFEST + ".tmp%d" % os.getpid(), "w") as f:
            f.write(source_hash() + "\n")
        os.replace(MANIFEST + ".tmp%d" % os.getpid(), MANIFEST)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
