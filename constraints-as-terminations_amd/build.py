"""In-tree build of libcatppo.so (hipcc, gfx950 only) and of the C oracle (gcc).

    python constraints-as-terminations_amd/build.py [--force]

Outputs (git-ignored, they travel to the GPU box with the gpurun snapshot):
    constraints-as-terminations_amd/lib/libcatppo.so
    oracle/liboracle_c.so
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libcatppo.so")
ORACLE_C = os.path.join(ROOT, "oracle", "c_oracle.c")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle_c.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the CaT / GAE / normaliser kernels must round every product and sum
# separately (bit-exact parity with the reference's separate eager kernels).  The GEMM kernels
# issue their FMAs explicitly through MFMA builtins, so the flag costs them nothing.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return r.stdout + r.stderr


def build_hip(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "catppo.h"))
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s[:-4] + ".o")
        if force or not _newer(obj, [src, *hdrs]):
            jobs.append([HIPCC, *HIP_FLAGS, "-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    objs = [os.path.join(OBJDIR, s[:-4] + ".o") for s in srcs]
    if force or jobs or not _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


def build_variant(name: str, extra_flags, verbose: bool = False) -> str:
    """A/B builds of the library (tools/gpu_ab.sh, CATPPO_LIB): the same sources with extra compile flags, e.g.
    ``build_variant("nopipe", ["-DGEMM_PIPE=0"])`` -> tools/bin/libcatppo_nopipe.so (git-ignored, travels with gpurun)"""
    objdir = os.path.join(OBJDIR, "variant_" + name)
    outdir = os.path.join(ROOT, "tools", "bin")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(outdir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = [[HIPCC, *HIP_FLAGS, *extra_flags, "-c", os.path.join(CSRC, s), "-o", os.path.join(objdir, s[:-4] + ".o")]
            for s in srcs]
    with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    lib = os.path.join(outdir, f"libcatppo_{name}.so")
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib,
          *[os.path.join(objdir, s[:-4] + ".o") for s in srcs]])
    return lib


def build_oracle_c(force: bool = False) -> str | None:
    if not os.path.exists(ORACLE_C):
        return None
    if force or not _newer(ORACLE_LIB, [ORACLE_C]):
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
              "-o", ORACLE_LIB, ORACLE_C, "-lm"])
    return ORACLE_LIB


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--variant":       # build.py --variant NAME [flags...]
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
        sys.exit(0)
    force = "--force" in sys.argv
    print(build_hip(force))
    print(build_oracle_c(force))
