"""Build libvog_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python vognet-pytorch_amd/csrc/build.py [--force]

Objects are rebuilt only when their source (or a header) is newer. The .so is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRCS = ["forward.hip", "gemm.hip", "attention.hip", "elementwise.hip", "lstm.hip", "txtail.hip", "visenc.hip", "pair.hip", "loss.hip", "assemble.hip", "backward.hip"]
HDRS = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")) + [os.path.join(ROOT, "include", "vog_hip.h")]
OUT = os.path.join(HERE, "libvog_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _cc(src):
    obj = os.path.join(HERE, src.replace(".hip", ".o"))
    path = os.path.join(HERE, src)
    if _stale(obj, [path] + HDRS):
        cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    if force:
        for s in SRCS:
            o = os.path.join(HERE, s.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=len(SRCS)) as ex:
        objs = list(ex.map(_cc, SRCS))
    if _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
