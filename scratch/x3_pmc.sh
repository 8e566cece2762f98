#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3/x3pmc
mkdir -p $O
for w in conv gemm; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_$w -- python3 $R/scratch/x3_gemm_only.py $w > /dev/null 2>&1
cp $(find /tmp/pmc_$w -name "*counter_collection.csv" | head -1) $O/sq_$w.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc2_$w -- python3 $R/scratch/x3_gemm_only.py $w > /dev/null 2>&1
cp $(find /tmp/pmc2_$w -name "*counter_collection.csv" | head -1) $O/sq2_$w.csv
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/r3/x3pmc/*.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm_x3_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k: sum(v) / len(v) for k, v in agg.items()})
PY
