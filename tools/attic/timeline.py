"""Tuning aid: summarise a rocprofv3 --kernel-trace csv of `bench.py --no-extras`:
per-queue busy time, stage-A queue gaps, whole-GPU idle time per step."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# step marker: a kernel that runs exactly once per step on the stage-A queue
conv = [r for r in rows if "k_wino_f23_input" in r["Kernel_Name"]]
if not conv:   # conv_algo="direct": the FPN convolution itself
    conv = [r for r in rows if "k_gemm_tile<64, 64, 32, 32, 2" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 200000]
q = collections.Counter(r["Queue_Id"] for r in conv).most_common(1)[0][0]
cq = [r for r in conv if r["Queue_Id"] == q]
a, b = 12, 44
t0, t1, steps = cq[a]["s"], cq[b]["s"], b - a
print("A queue %s, %d steps, wall %.3f ms/step" % (q, steps, (t1 - t0) / steps / 1e6))
byq = collections.defaultdict(list)
for r in rows:
    if r["s"] >= t0 and r["e"] <= t1:
        byq[r["Queue_Id"]].append(r)
for qq, rs in sorted(byq.items()):
    print("queue %s: %.1f kernels/step, busy %.3f ms/step" % (qq, len(rs) / steps, sum(r["e"] - r["s"] for r in rs) / steps / 1e6))
ev = []
for rs in byq.values():
    for r in rs:
        ev += [(r["s"], 1), (r["e"], -1)]
ev.sort()
depth, last, idle, multi = 0, t0, 0, 0
for t, d in ev:
    if depth == 0: idle += t - last
    if depth >= 2: multi += t - last
    depth += d; last = t
print("GPU idle %.3f ms/step; >= 2 kernels in flight %.3f ms/step" % (idle / steps / 1e6, multi / steps / 1e6))
gaps, prev, agg = 0, None, collections.defaultdict(lambda: [0, 0])
for r in byq[q]:
    agg[r["Kernel_Name"][:46]][0] += 1; agg[r["Kernel_Name"][:46]][1] += r["e"] - r["s"]
    if prev is not None: gaps += max(0, r["s"] - prev)
    prev = max(prev or 0, r["e"])
print("A queue: gaps between consecutive kernels %.3f ms/step" % (gaps / steps / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-48s %5.1f/step %8.1f us avg %7.3f ms/step" % (k, v[0] / steps, v[1] / v[0] / 1e3, v[1] / steps / 1e6))
