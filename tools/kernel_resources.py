"""Compiles one .hip source of the library with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs, spills, scratch, occupancy, LDS, code size. python tools/kernel_resources.py [pt_kernels.hip] [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import build as B

src = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "pt_kernels.hip"
extra = [a for a in sys.argv[1:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    obj = os.path.join(d, "o.o")
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, src), "-I", B.CSRC, "-o", obj,
                                                      "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        print(res.stdout[-4000:])
        sys.exit(1)
    cur = None
    rows = {}
    for line in res.stdout.splitlines():
        m = re.search(r"remark:\s+(.*?)(?: \[-Rpass)", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    dem = subprocess.run(["c++filt"] + list(rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>6s}")
    for (name, r), dn in zip(rows.items(), dem):
        dn = re.sub(r"^void akr::", "", dn).replace("(akr::PtParams)", "")
        print(f"{dn[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} "
              f"{r.get('SGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>3s} {r.get('LDS Size [bytes/block]', '?'):>6s}")
