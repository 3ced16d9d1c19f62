"""Which hardware queue each stream's kernels went to, from a rocprofv3 --kernel-trace csv:
   python tools/queue_map.py <..._kernel_trace.csv>
One line per (stream, queue) pair with the number of kernels and the time from its first start to its last end."""
import csv, sys
from collections import defaultdict
rows = defaultdict(lambda: [0, None, None])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if not any(k in name for k in ("mfcc_base", "finish_kernel", "infer_vote")): continue
        key = (int(r["Stream_Id"]), int(r["Queue_Id"]))
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        v = rows[key]
        v[0] += 1
        v[1] = s if v[1] is None else min(v[1], s)
        v[2] = e if v[2] is None else max(v[2], e)
t0 = min(v[1] for v in rows.values())
for (stream, queue), (n, s, e) in sorted(rows.items(), key=lambda kv: kv[1][1]):
    print("stream %3d  queue %2d  kernels %6d  from %9.3f ms to %9.3f ms" % (stream, queue, n, (s - t0) / 1e6, (e - t0) / 1e6))
