"""Register / scratch / occupancy report of the per-scene kernels (host/specialise.cpp) of the textured room, next to what the
precompiled interpreter kernels of the same instantiation need: compiles the generated text with hipcc
-Rpass-analysis=kernel-resource-usage (no GPU needed). python tools/spec_resources.py [n_floor=1] [--keep DIR]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import build as B, capi
from tests.helpers import textured_room


def report(header: str, variants, extra=()):
    """variants: [(label, bvh, pmj, stage, defer, waves, spec)] -> rows of the compiler's resource remarks"""
    rows = []
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "akr_scene_spec.h"), "w").write(header)
        for label, bvh, pmj, stage, defer, waves, spec in variants:
            b = lambda x: "true" if x else "false"  # noqa: E731
            absent = "akr::kSpecAbsent" if spec else "0u"
            src = (("#define AKR_SPEC_GRAPHS 1\n" if spec else "") + '#include "device/pt_pass.h"\n'
                   f'extern "C" __global__ __launch_bounds__(256, {waves}) void k(const akr::PtParams p) {{\n'
                   f"    akr::pt_pass_body<{b(bvh)}, false, true, {b(pmj)}, {b(stage)}, {b(defer)}, {absent}>(p);\n}}\n")
            path = os.path.join(d, "k.hip")
            open(path, "w").write(src)
            cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + list(extra) + ["-c", path, "-I", B.CSRC, "-I", d, "-o", os.path.join(d, "k.o"),
                                                                      "-Rpass-analysis=kernel-resource-usage"]
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if res.returncode != 0:
                raise RuntimeError(res.stdout[-3000:])
            r = {}
            for line in res.stdout.splitlines():
                m = re.search(r"remark:\s+(.*?)(?: \[-Rpass)", line)
                if m and ":" in m.group(1):
                    k, v = m.group(1).split(":", 1)
                    r[k.strip()] = v.strip()
            rows.append((label, r))
    return rows


if __name__ == "__main__":
    n_floor = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
    sc = capi.Scene(None, textured_room(64, 64, n_floor=n_floor))
    header = sc.spec_source()
    bvh = n_floor > 1
    variants = []
    for spec in (False, True):
        for waves in (3, 4):
            variants.append((f"{'per-scene' if spec else 'interpreter'} {'bvh' if bvh else 'exhaustive'} waves={waves}", bvh, False, True, True, waves, spec))
    print(f"{'kernel':44s} {'VGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>3s}")
    for label, r in report(header, variants):
        print(f"{label:44s} {r.get('VGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>3s}")
