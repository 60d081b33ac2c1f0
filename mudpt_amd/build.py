"""Build libmudpt_hip.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "lib", "libmudpt_hip.so")
SOURCES = ["gemm.hip", "gemm_pp.hip", "attention.hip", "attention_single.hip", "attention_exact.hip", "attention_resident.hip", "layernorm.hip", "elementwise.hip", "head.hip", "model.cpp"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "mudpt.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
EXTRA_FLAGS = {"attention_resident.hip": ["-fno-slp-vectorize"]}  # why: the header of that file


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel / host sources the library is built from: stamped into profiles/*_bytes_per_step.json by
    tools/profile_round.sh, compared by bench.py -- a traffic figure measured on other kernels than the ones loaded is reported as null."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + ["common.h", "kernels.h"]):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append([cc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn, file=sys.stderr)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
