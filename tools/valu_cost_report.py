"""Static issue-cost attribution of a stretch of gfx950 assembly (hipcc -S --cuda-device-only), priced with the per-instruction
costs measured by tools/micro/vgpr_bank.hip and tools/micro/valu_rate.hip on MI355X at 4 waves per SIMD
(profiles/r4_vgpr_bank_*.txt). PC sampling / thread trace are not available on this lease (profiles/r4_pc_sampling_unavailable.txt),
so "which instructions the waves wait on" is answered the other way round: what every instruction of the hot loop costs its SIMD.

Cost table (cycles of one SIMD per wave64 instruction):
  2-source VALU (mul, add, sub, min, max, fmac-free ...)                  2.2
  3-VGPR-source VALU (fma / fmac / mad ...)                               2.2 (VOP2 fmac) - 2.6 (VOP3), priced 2.3
    ... whose three source VGPRs all have the SAME PARITY                 4.4   <- the register file's two banks (even / odd): "bank"
    ... with an SGPR source                                               4.4   <- "sgpr"
  v_cmp_* (to vcc or to an SGPR pair), v_cndmask_b32_e64                  4.4
  v_cndmask_b32_e32 (vcc)                                                 3.7
  v_min3 / v_max3 / v_med3, v_bfi, v_div_scale / _fmas / _fixup           4.4
  v_rcp / v_rsq / v_sqrt / v_log / v_exp (quarter rate)                   8.2
  v_pk_{fma,mul,add}_f32 (two results)                                    4.4
  v_max_f32 x, x (the compiler's canonicalisation of a possible sNaN)     4.4
usage: valu_cost_report.py file.s first_line last_line [first last ...] [--json out.json] [--records n]"""
import json
import re
import sys

VREG = re.compile(r"^-?\|?v(\d+)\|?$")
SREG = re.compile(r"^-?\|?(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0)\|?$")
THREE = {"v_fma_f32", "v_mad_f32", "v_mad_u32_u24", "v_mad_i32_i24", "v_lshl_add_u32", "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_xad_u32",
         "v_bfe_u32", "v_bfe_i32", "v_alignbit_b32", "v_perm_b32", "v_add_lshl_u32", "v_fma_f16", "v_lshl_add_u64"}
DOUBLE = {"v_min3_f32", "v_max3_f32", "v_med3_f32", "v_min3_u32", "v_max3_u32", "v_med3_u32", "v_min3_i32", "v_max3_i32", "v_bfi_b32", "v_div_scale_f32", "v_div_fmas_f32",
          "v_div_fixup_f32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_exp_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    ops = [o.strip().split(" ")[0] for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def vnum(op):
    m = VREG.match(op)
    return int(m.group(1)) if m else None


def price(mn, ops):
    """(cycles, class) of one instruction; non-VALU instructions cost the VALU nothing here (class 'other')."""
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if not base.startswith("v_"):
        return 0.0, "salu / branch / waitcnt" if base.startswith("s_") else ("lds" if base.startswith("ds_") else "vmem")
    if base.startswith("v_pk_"):
        return 4.4, "packed f32 (two results)"
    if base in TRANS:
        return 8.2, "transcendental (rcp ...)"
    if base.startswith("v_cmp"):
        return 4.4, "compare"
    if base == "v_cndmask_b32":
        return (4.4, "cndmask (SGPR-pair mask)") if mn.endswith("_e64") else (3.7, "cndmask (vcc)")
    if base in DOUBLE:
        return 4.4, "double-rate op (min3 / div_* / 32-bit mul)"
    if base == "v_max_f32" and len(ops) >= 3 and ops[1] == ops[2]:
        return 4.4, "canonicalise (v_max x, x)"
    src = None
    if base in ("v_fmac_f32", "v_mac_f32") and len(ops) >= 3:
        src = (ops[1], ops[2], ops[0])
    elif base in THREE and len(ops) >= 4:
        src = (ops[1], ops[2], ops[3])
    if src:
        nums = [vnum(s) for s in src]
        if any(SREG.match(s) for s in src):
            return 4.4, "3-source op with an SGPR source"
        if all(n is not None for n in nums) and len({n % 2 for n in nums}) == 1 and len(set(nums)) > 1:
            return 4.4, "3-VGPR-source op, all sources in one register bank (same parity)"
        return 2.3, "3-source op, conflict-free"
    return 2.2, "2-source / 1-source VALU"


def main():
    args = sys.argv[1:]
    out_json = None
    records = None
    if "--json" in args:
        i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
    if "--records" in args:
        i = args.index("--records"); records = float(args[i + 1]); del args[i:i + 2]
    path = args[0]
    lines = open(path).read().splitlines()
    ranges = [(int(args[i]), int(args[i + 1])) for i in range(1, len(args) - 1, 2)] or [(1, len(lines))]
    classes = {}
    per_inst = []
    n_valu = 0
    for lo, hi in ranges:
        for ln_no in range(lo - 1, hi):
            p = parse(lines[ln_no])
            if not p:
                continue
            cyc, cls = price(*p)
            c = classes.setdefault(cls, {"instructions": 0, "cycles": 0.0})
            c["instructions"] += 1
            c["cycles"] += cyc
            if p[0].startswith("v_"):
                n_valu += 1
            per_inst.append((cyc, ln_no + 1, lines[ln_no].split(";")[0].strip()))
    total = sum(c["cycles"] for c in classes.values())
    ideal = 2.2 * n_valu
    print(f"{path}: lines {ranges}: {n_valu} VALU instructions, modelled {total:.0f} SIMD cycles ({total / max(1, n_valu):.2f} per instruction; "
          f"{ideal:.0f} if every one cost 2.2)")
    if records:
        print(f"  = {total / records:.0f} cycles per record of the walk")
    for cls, c in sorted(classes.items(), key=lambda kv: -kv[1]["cycles"]):
        print(f"  {c['cycles']:8.1f} cycles {100 * c['cycles'] / max(1e-9, total):5.1f} %  {c['instructions']:4d} x  {cls}")
    if out_json:
        json.dump({"file": path, "ranges": ranges, "valu_instructions": n_valu, "modelled_cycles": total, "cycles_if_all_2.2": ideal,
                   "classes": classes, "instructions": [{"cycles": c, "line": n, "text": t} for c, n, t in per_inst]}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
