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
    "host/image_formats.cpp",
    "host/pmj_tables.cpp",
    "host/comm.cpp",
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
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
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


OBJ = os.path.join(HERE, "build", "obj")  # git-ignored; objects are rebuilt per source file, in parallel


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        build_cli(verbose)
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("AKR_EXTRA_HIPCC_FLAGS", "").split()
    os.makedirs(OBJ, exist_ok=True)
    tag = os.path.join(OBJ, "flags.txt")  # objects built with other flags are stale
    flags_now = " ".join(FLAGS + extra)
    if force or not os.path.exists(tag) or open(tag).read() != flags_now:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    newest_header = max([os.path.getmtime(os.path.normpath(os.path.join(CSRC, h))) for h in _headers()] + [os.path.getmtime(os.path.abspath(__file__))])

    def compile_one(src: str):
        obj = os.path.join(OBJ, src.replace("/", "_") + ".o")
        path = os.path.join(CSRC, src)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj, 0, ""
        cmd = [hipcc] + FLAGS + extra + ["-c", path, "-I", CSRC, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return obj, res.returncode, res.stdout

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    log = "".join(out for _, _, out in results)
    if any(rc != 0 for _, rc, _ in results):
        raise RuntimeError("hipcc failed:\n" + log)
    open(tag, "w").write(flags_now)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _, _ in results] + ["-ldl", "-o", LIB]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + log + res.stdout)
    if verbose and (log + res.stdout).strip():
        print(log + res.stdout, file=sys.stderr)
    build_cli(verbose)
    return LIB


def build_variant(name: str, extra_flags, verbose: bool = False, only=None) -> str:
    """An A/B build of the library with extra compiler flags (e.g. -DAKR_BVH_NODE_WORDS=32) next to the product:
    akari_render_amd/variants/libakari_hip_<name>.so, selected at run time with AKR_HIP_LIB=<path> (measurement only).
    only = the sources the flags concern (e.g. ["pt_kernels.hip"]): the others are linked from the product's objects."""
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(HERE, "build", "obj_" + name)
    out_dir = os.path.join(HERE, "variants")
    os.makedirs(obj, exist_ok=True)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"libakari_hip_{name}.so")
    if only:
        build(verbose=verbose)  # the product's objects must be current

    def compile_one(src: str):
        if only and src not in only:
            return os.path.join(OBJ, src.replace("/", "_") + ".o"), 0, ""
        o = os.path.join(obj, src.replace("/", "_") + ".o")
        res = subprocess.run([hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-I", CSRC, "-o", o],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return o, res.returncode, res.stdout

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    if any(rc != 0 for _, rc, _ in results):
        raise RuntimeError("hipcc failed:\n" + "".join(o for _, _, o in results))
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _, _ in results] + ["-ldl", "-o", out],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # --variant NAME [--only a.hip,b.cpp] FLAGS...
        rest = sys.argv[3:]
        only = None
        if rest and rest[0] == "--only":
            only, rest = rest[1].split(","), rest[2:]
        print(build_variant(sys.argv[2], rest, verbose=True, only=only))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
