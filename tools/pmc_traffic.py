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
    if m:   # <BM, BN, WM, WN, AMODE, positional add>
        return "k_gemm_tile<%s,%s,%s,%s,%s>" % (m.group(1), m.group(2), m.group(3), m.group(4),
                                                 AMODE.get(m.group(5), "A_STEM"))
    if name.startswith("void k_attn_small") or name.startswith("k_attn_chunk"):
        return "k_attn_chunk"     # (bench.py reports both attention kernels under this name)
    if name.startswith("void k_gemm_s3<true>"):
        return "k_gemm_s3<ln>"    # (bench.py: the row-epilogue instantiation)
    if name.startswith(("void k_gemm_s3<false>", "k_gemm_s3_wide", "k_gemm_s3_narrow")):
        return "k_gemm_s3"        # (bench.py reports the plain-epilogue tile forms under one name)
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


def sq_counters(root):
    """kernel -> counter -> [launches, sum] from the SQ pass."""
    acc = {}
    for path in glob.glob(os.path.join(root, "mfma", "**", "*counter_collection.csv"),
                          recursive=True):
        for r in csv.DictReader(open(path)):
            a = acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc


def main(root):
    fetch, write = counters(root, "fetch"), counters(root, "write")
    sq = sq_counters(root)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counters only) on "
                     "`bench.py --steps 3 --warmup 2 --no-pipeline --no-graphs`, MI355X; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of "
                     "wide coalesced reads); KB = 1024 B.  Kernels that share a bench_name "
                     "(template variants) are merged by bench.py weighted by launches.  `sq`: a "
                     "third pass with SQ counters; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / "
                     "(4 x SQ_BUSY_CU_CYCLES), the fraction of matrix-pipe cycles busy while the "
                     "CU is busy.",
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
        c = sq.get(k)
        if c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"][1] > 0:
            per = lambda name: c[name][1] / c[name][0] if name in c else None
            busy, cu = per("SQ_VALU_MFMA_BUSY_CYCLES"), per("SQ_BUSY_CU_CYCLES")
            wave = per("SQ_WAVE_CYCLES")
            out["kernels"][k]["sq"] = {
                "mfma_busy_cycles": busy, "busy_cu_cycles": cu,
                # 4 SIMDs (matrix pipes) per CU
                "mfma_util": round(busy / (4.0 * cu), 4) if busy is not None else None,
                "wait_inst_any_frac": round(per("SQ_WAIT_INST_ANY") / wave, 4) if wave else None,
                "active_inst_any_frac": round(per("SQ_ACTIVE_INST_ANY") / wave, 4) if wave else None,
                "lds_bank_conflict_cycles": per("SQ_LDS_BANK_CONFLICT"),
                "avg_us": round(c["SQ_BUSY_CU_CYCLES"][2] / c["SQ_BUSY_CU_CYCLES"][0] / 1e3, 2)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
