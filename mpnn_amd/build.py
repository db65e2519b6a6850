"""Build libmpnn_amd.so (the C-ABI library) in-tree with hipcc for gfx950.

    python -m mpnn_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the source snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmpnn_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-fno-fast-math"] + os.environ.get("MPNN_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "mpnn_amd.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def _compile_one(job):
    hipcc, src, obj, verbose = job
    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    return src, r.returncode, r.stdout + r.stderr


def build(force=False, verbose=False):
    """Compile every HIP source to an object (only the stale ones, a few in parallel) and link them into one shared
    object; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmpnn_amd.so (and there is no CPU fallback)")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "mpnn_amd.h")]
    t_hdr = max(os.path.getmtime(p) for p in headers)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_hdr):
            jobs.append((hipcc, src, obj, verbose))
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("MPNN_BUILD_JOBS", "6")))) as ex:
            for src, rc, log in ex.map(_compile_one, jobs):
                if rc != 0:
                    sys.stderr.write(log)
                    raise RuntimeError("hipcc failed on %s" % src)
                if verbose and log:
                    sys.stderr.write(log)
    tmp = LIB + ".tmp"
    r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libmpnn_amd.so")
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
