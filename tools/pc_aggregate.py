"""Aggregates a rocprofv3 PC-sampling CSV (host_trap or stochastic) of one bench.py step into per-instruction / per-source-line
shares for the k_pt_pass dispatches.  usage: pc_aggregate.py <pc_sampling.csv> <kernel_trace.csv> <out prefix>
Writes <prefix>.json and <prefix>.txt (top PCs with the tool's disassembly text and source comment, share of samples; for
stochastic sampling also the stall reason and whether the wave issued)."""
import collections
import csv
import json
import sys


def main(pcs, ktrace, prefix, top=40):
    kernels = {}
    if ktrace:
        for row in csv.DictReader(open(ktrace)):
            kernels[row.get("Dispatch_Id")] = row.get("Kernel_Name", "")
    by_inst = collections.Counter()
    by_line = collections.Counter()
    by_type = collections.Counter()
    by_reason = collections.Counter()
    issued = collections.Counter()
    reason_of_inst = collections.defaultdict(collections.Counter)
    lanes = collections.Counter()
    n = 0
    n_other = 0
    cols = None
    for row in csv.DictReader(open(pcs)):
        cols = cols or list(row.keys())
        k = kernels.get(row.get("Dispatch_Id"), "k_pt_pass" if not kernels else "")
        if "k_pt_pass" not in k:
            n_other += 1
            continue
        n += 1
        inst = row.get("Instruction", "?").strip()
        cm = row.get("Instruction_Comment", "").strip()
        key = (inst, cm)
        by_inst[key] += 1
        by_line[cm.split(" ")[0] if cm else "?"] += 1
        em = row.get("Exec_Mask")
        if em:
            try:
                lanes[key] += bin(int(em)).count("1")
            except ValueError:
                pass
        if "Stall_Reason" in row:
            r = row.get("Stall_Reason", "")
            by_reason[r] += 1
            reason_of_inst[key][r] += 1
            issued[row.get("Wave_Issued_Instruction", "")] += 1
            by_type[row.get("Instruction_Type", "")] += 1
    res = {"source_csv": pcs, "columns": cols, "samples_in_k_pt_pass": n, "samples_elsewhere": n_other,
           "kernel": sorted({v.split("(")[0] for v in kernels.values() if "k_pt_pass" in v}),
           "top_instructions": [{"instruction": i, "source": c, "samples": v, "share": v / max(1, n), "mean_active_lanes": (lanes[(i, c)] / v if lanes[(i, c)] else None),
                                 "stall_reasons": dict(reason_of_inst[(i, c)].most_common(4)) if reason_of_inst else None}
                                for (i, c), v in by_inst.most_common(top)],
           "top_source_lines": [{"line": l, "samples": v, "share": v / max(1, n)} for l, v in by_line.most_common(top)]}
    if by_reason:
        res["stall_reason_shares"] = {k: v / n for k, v in by_reason.most_common()}
        res["wave_issued_shares"] = {k: v / n for k, v in issued.most_common()}
        res["instruction_type_shares"] = {k: v / n for k, v in by_type.most_common()}
    # by mnemonic class
    cls = collections.Counter()
    for (i, _), v in by_inst.items():
        cls[i.split(" ")[0]] += v
    res["by_mnemonic"] = [{"mnemonic": m, "share": v / max(1, n)} for m, v in cls.most_common(25)]
    json.dump(res, open(prefix + ".json", "w"), indent=1)
    with open(prefix + ".txt", "w") as f:
        f.write(f"{n} samples in k_pt_pass ({n_other} elsewhere); kernel {res['kernel']}\n")
        if by_reason:
            f.write("stall reasons: " + ", ".join(f"{k} {v:.3f}" for k, v in res["stall_reason_shares"].items()) + "\n")
            f.write("wave issued:   " + ", ".join(f"{k} {v:.3f}" for k, v in res["wave_issued_shares"].items()) + "\n")
            f.write("inst types:    " + ", ".join(f"{k} {v:.3f}" for k, v in res["instruction_type_shares"].items()) + "\n")
        f.write("by mnemonic:   " + ", ".join(f"{d['mnemonic']} {d['share']:.3f}" for d in res["by_mnemonic"]) + "\n\n")
        f.write("top source lines\n")
        for d in res["top_source_lines"][:25]:
            f.write(f"  {d['share']*100:6.2f} %  {d['line']}\n")
        f.write("\ntop instructions\n")
        for d in res["top_instructions"]:
            f.write(f"  {d['share']*100:6.2f} %  {d['instruction']:<60s} {d['source']}  {d['stall_reasons'] or ''}\n")
    print(open(prefix + ".txt").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3] if len(sys.argv) > 3 else "pc_summary")
