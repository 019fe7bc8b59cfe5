"""Steady-state timeline of the pipelined bench from a rocprofv3 --kernel-trace csv: how much of
a step the chip spends with 0 / 1 / 2 chip-filling MFMA kernels in flight."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"]
rows.sort(key=lambda r: r["s"])
marks = [r for r in rows if "k_msda" in r["n"]][::6]      # one mark per image
a, b = len(marks) - 90, len(marks) - 10                    # inside the final timed loop
t0, t1, steps = marks[a]["s"], marks[b]["s"], b - a
print("%d steps, wall %.3f ms/step" % (steps, (t1 - t0) / steps / 1e6))
win = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
big = lambda r: ("k_gemm_tile" in r["n"] or "k_gemm_group" in r["n"] or "k_gemm_rowln" in r["n"])
def union(rs):
    ev = sorted([(r["s"], 1) for r in rs] + [(r["e"], -1) for r in rs])
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        depth += d; last = t
    hist[0] += t1 - last
    return hist
h = union([r for r in win if big(r)])
print("MFMA tile kernels in flight: " + ", ".join("%d: %.3f ms" % (k, h[k] / steps / 1e6) for k in sorted(h)))
print("sum of their durations %.3f ms/step" % (sum(r["e"] - r["s"] for r in win if big(r)) / steps / 1e6))
h = union(win)
print("any kernel in flight: " + ", ".join("%d: %.3f ms" % (k, h[k] / steps / 1e6) for k in sorted(h)))
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    k = r["n"][:60]
    agg[k][0] += 1; agg[k][1] += r["e"] - r["s"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print("  %-62s %6.1f/step %8.1f us avg %7.3f ms/step" % (k, v[0] / steps, v[1] / v[0] / 1e3, v[1] / steps / 1e6))
