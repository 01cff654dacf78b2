"""Build recipe for libggad_hip.so (gfx950 only, in-tree).

    python -m ggad_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU present.  The shared object lands next to this
file (git-ignored, but it travels to the GPU box with the work-tree snapshot).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libggad_hip.so")
STAMP = os.path.join(HERE, ".libggad_hip.stamp")
SOURCES = ["runtime.cpp", "plan_build.cpp", "exchange.cpp", "sampler.cpp", "sampler_x86.cpp", "spmm_panel_build.cpp", "spmm_ring_build.cpp", "plan.hip", "hop2_ldsw.hip", "step.hip", "step_xcd.hip", "fullgraph.hip", "gemm.hip", "gemm_slab.hip", "mlp.hip", "baselines.hip"]
HOST_ONLY = {"sampler.cpp", "sampler_x86.cpp", "spmm_panel_build.cpp", "spmm_ring_build.cpp"}      # plain C++ (x86 intrinsics behind a run-time CPU check), no device pass
HEADERS = ["common.h", "step_common.h", "libggad_hip.map", os.path.join("..", "..", "include", "ggad_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"] + os.environ.get("GGAD_EXTRA_HIPFLAGS", "").split()      # e.g. -DGGAD_G2_PROF (scripts/g2_phase_clocks.py)


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; libggad_hip.so cannot be built")


def _digest(srcs) -> str:
    h = hashlib.sha256()
    for name in list(srcs) + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    dig = _digest(srcs)
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in srcs:
        obj = os.path.join(bdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if s in HOST_ONLY:
            cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-Wall", "-x", "c++", "-c", os.path.join(CSRC, s), "-o", obj]
        else:
            cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print("[ggad build]", " ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "libggad_hip.map"), "-o", LIB] + objs
    if verbose:
        print("[ggad build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
