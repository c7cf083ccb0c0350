"""Builds libmi355x_asr.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m nemo_amd.build            # incremental (mtime based)
    python -m nemo_amd.build --force

hipcc cross-compiles without a GPU; the resulting nemo_amd/lib/libmi355x_asr.so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmi355x_asr.so")
OBJDIR = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I", INCLUDE, "-I", CSRC,
         "-Wno-unused-result"] + os.environ.get("MI355X_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))] + [
        os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    jobs = []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src[:-4] + ".o")
        if force or _newer(s, o) or any(_newer(h, o) for h in hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    failed = False
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for s, rc, out in ex.map(compile_one, jobs):
            if verbose:
                print(f"[build] hipcc {os.path.basename(s)} -> rc={rc}")
            if rc != 0:
                failed = True
                print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(OBJDIR, src[:-4] + ".o") for src in sources()]
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
