"""Reads a rocprofv3 --kernel-trace CSV of a wavefront render and prints where the time of the LAST launch group went: span, time inside kernels per
kernel name, idle time between kernels, and the duration of every n-th trace launch.  python tools/wf_timeline.py <kernel_trace.csv>"""
import csv, sys, json
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last launch group: from the last k_wf_init on
inits = [i for i, r in enumerate(rows) if "k_wf_init" in r[2]]
if not inits:
    sys.exit("no k_wf_init in the trace")
g = rows[inits[-1]:]
t0, t1 = g[0][0], max(r[1] for r in g)
by = {}
for s, e, n in g:
    k = n.split("(")[0].replace("void akr::", "")[:40]
    by.setdefault(k, [0, 0])
    by[k][0] += 1; by[k][1] += e - s
busy = 0; cur_s, cur_e = g[0][0], g[0][1]
for s, e, n in g[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tr = [(s, e) for s, e, n in g if "k_wf_trace" in n]
sh = [(s, e) for s, e, n in g if "k_wf_shade" in n]
print(json.dumps({"span_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6, "kernels": {k: {"n": v[0], "ms": v[1] / 1e6} for k, v in by.items()}}))
step = max(1, len(tr) // 24)
print("trace launch: duration us", [round((e - s) / 1e3) for s, e in tr[::step]])
print("shade launch: duration us", [round((e - s) / 1e3) for s, e in sh[::step]])
