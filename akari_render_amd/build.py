"""Builds libakari_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libakari_hip.so")

SOURCES = [
    "pt_kernels.hip",
    "wf_kernels.hip",
    "aov_kernels.hip", "gpt_kernels.hip", "mcmc_kernels.hip",
    "host/api.cpp",
    "host/scene_build.cpp",
    "host/scene_json.cpp",
    "host/bvh.cpp",
    "host/image_io.cpp",
    "host/pmj_tables.cpp",
]
def _headers():
    """Every header the sources can include: all of csrc/ and the public header (a forgotten entry in a hand-kept list
    means a stale library that still loads)."""
    out = []
    for d, _, files in os.walk(CSRC):
        out += [os.path.relpath(os.path.join(d, f), CSRC) for f in files if f.endswith(".h")]
    return sorted(out) + ["../../include/akari_hip.h"]


# -ffp-contract=off + correctly rounded div/sqrt: the arithmetic contract of DESIGN.md ("AKR-F32")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    # packed-f32 SLP vectorisation costs s_mov shuffles of the SGPR triangle records (-5 % on the bench)
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function", "-x", "hip",
]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + _headers() + ["../build.py"]:
        p = os.path.normpath(os.path.join(CSRC, f))
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


CLI = os.path.join(HERE, "akari-cli")


def build_cli(verbose: bool = False) -> str:
    """akari-cli: the reference's command line (crates/akari_api/src/bin/akari_cli.rs) over libakari_hip.so."""
    src = os.path.join(CSRC, "host", "cli_main.cpp")
    if os.path.exists(CLI) and os.path.getmtime(CLI) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", src, "-I", os.path.join(HERE, "..", "include"), "-L", HERE, "-lakari_hip",
           "-Wl,-rpath,$ORIGIN", "-o", CLI]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stdout)
    return CLI


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        build_cli(verbose)
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("AKR_EXTRA_HIPCC_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-I", CSRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    if verbose and res.stdout.strip():
        print(res.stdout, file=sys.stderr)
    build_cli(verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
