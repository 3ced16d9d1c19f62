"""Summary of a rocprofv3 --kernel-trace CSV: per kernel count / average duration, and for the busiest
window of the run the wall time, the union of kernel intervals (GPU busy) and the sum of durations.
usage: python tools/trace_summary.py <dir with *_kernel_trace.csv> [name filter]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    per[n][0] += 1; per[n][1] += e - s
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{c:8d} x {t / c / 1e3:9.2f} us  total {t / 1e6:9.2f} ms  {n[:110]}")
flt = sys.argv[2] if len(sys.argv) > 2 else None
sel = [(s, e) for s, e, n in rows if flt is None or flt in n]
if sel:
    t0, t1 = sel[0][0], max(e for _, e in sel)
    inside = [(s, e) for s, e, n in rows if s >= t0 and e <= t1]
    busy, cur_s, cur_e = 0, None, None
    for s, e in inside:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"span {(t1 - t0) / 1e6:.2f} ms, GPU busy (union) {busy / 1e6:.2f} ms, sum of durations {sum(e - s for s, e in inside) / 1e6:.2f} ms, kernels {len(inside)}")
