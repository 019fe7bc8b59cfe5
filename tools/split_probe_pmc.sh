#!/bin/bash
# Counter passes (rocprofv3 --pmc only) over tools/bin/gemm_split_probe on ONE shape; per-kernel
# means go to gpurun_out/split_probe_pmc.txt.   bash tools/split_probe_pmc.sh [shape index]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
SH=${1:-0}; OUT=gpurun_out/split_pmc; rm -rf $OUT; mkdir -p $OUT
G=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
   "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
   "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM"
   "SQ_INST_LEVEL_LDS SQ_INSTS_LDS"
   "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
   "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU")
i=0
for g in "${G[@]}"; do
  timeout 200 rocprofv3 --pmc $g --output-format csv -d $OUT/g$i -o p -- tools/bin/gemm_split_probe noverify $SH > $OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/split_pmc/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:44]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/split_probe_pmc.txt', 'w') as o:
    for k, d in acc.items():
        if 'split_w' in k: continue
        m = {c: sum(v) / len(v) for c, v in d.items()}
        o.write(k + '\n')
        wc = m.get('SQ_WAVE_CYCLES')
        for c in sorted(m):
            o.write('   %-30s %.4g' % (c, m[c]) + ('   /wave_cycles %.3f' % (m[c] / wc) if wc and c.startswith(('SQ_WAIT', 'SQ_ACTIVE')) else '') + '\n')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'SQ_BUSY_CU_CYCLES' in m:
            o.write('   mfma_busy %.3f\n' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES'])))
        if 'SQ_INST_LEVEL_VMEM' in m: o.write('   vmem_latency_cycles %.0f\n' % (m['SQ_INST_LEVEL_VMEM'] / m['SQ_INSTS_VMEM']))
        if 'SQ_INST_LEVEL_LDS' in m: o.write('   lds_latency_cycles %.0f\n' % (m['SQ_INST_LEVEL_LDS'] / m['SQ_INSTS_LDS']))
print(open('gpurun_out/split_probe_pmc.txt').read())
PY
