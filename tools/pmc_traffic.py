"""Fold the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into per-kernel HBM bytes.

FETCH_SIZE / WRITE_SIZE are reported in KB (1024 B); on gfx950 FETCH_SIZE counts half the
bytes of wide coalesced reads and is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import re
import sys

AMODE = {"0": "A_ROW", "1": "A_COL", "2": "A_CONV"}


def bench_name(name):
    m = re.match(r"void k_gemm_tile<(\d+), (\d+), (\d+), (\d+), (\d), (true|false)>", name)
    if m:
        return "k_gemm_tile<%s,%s,%s,%s,%s>" % (m.group(1), m.group(2), m.group(3), m.group(4),
                                                  AMODE[m.group(5)])
    m = re.match(r"void k_gemm_skinny<(\d), \d+>", name)
    if m:
        return "k_gemm_skinny<%s>" % AMODE[m.group(1)]
    return name.split("(")[0].replace("void ", "")


def counters(root, which):
    acc = {}
    for path in glob.glob(os.path.join(root, which, "**", "*counter_collection.csv"),
                          recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name", "").startswith(which.upper() + "_SIZE"):
                k = r["Kernel_Name"]
                a = acc.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc


def main(root):
    fetch, write = counters(root, "fetch"), counters(root, "write")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counters only) on "
                     "`bench.py --steps 3 --warmup 2 --no-pipeline --no-graphs`, MI355X; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of "
                     "wide coalesced reads); KB = 1024 B.  Kernels that share a bench_name "
                     "(template variants) are merged by bench.py weighted by launches.",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        fkb, wkb = (f / nf if nf else 0.0), (w / nw if nw else 0.0)
        out["kernels"][k] = {"launches_profiled": n, "fetch_size_KB_raw": round(fkb, 1),
                             "write_size_KB": round(wkb, 1),
                             "hbm_bytes_per_launch": int((2.0 * fkb + wkb) * 1024),
                             "bench_name": bench_name(k)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
