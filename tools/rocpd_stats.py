"""Per-kernel summary (the `--stats` table) from a rocprofv3 rocpd SQLite database.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import math
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, end - start from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        agg.setdefault(name, []).append(dur)
    total = float(sum(sum(v) for v in agg.values()))
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / n)
        print('"%s",%d,%d,%.6f,%.2f,%d,%d,%.6f' % (name, n, s, mean, 100.0 * s / total, min(v),
                                                    max(v), sd))


if __name__ == "__main__":
    main(sys.argv[1])
