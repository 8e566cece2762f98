#!/bin/bash
# counters of the mask-logit einsum kernels (scratch/x3_train_gemm_only.py), separate rocprofv3 --pmc passes -> gpurun_out/r6/x3_train_gemm_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r6
mkdir -p $O
: > $O/x3_train_gemm_pmc.txt
pass() { n=$1; shift
  rm -rf /tmp/xg_$n
  timeout 240 rocprofv3 --pmc "$@" --output-format csv -d /tmp/xg_$n -- python3 $R/scratch/x3_train_gemm_only.py > /dev/null 2>&1
  python3 $R/scratch/pmc_filter.py $(find /tmp/xg_$n -name "*counter_collection.csv" | head -1) cgg_gemm_x3_kernel > /tmp/xg_$n.csv
  python3 - /tmp/xg_$n.csv $n >> $O/x3_train_gemm_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r['Kernel_Name'].split('(')[0][:70], r.get('Grid_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), d in sorted(agg.items()):
    print('%s  %s grid %s: ' % (sys.argv[2], k, g) + ', '.join('%s %.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())) + '  (%d launches)' % max(len(v) for v in d.values()))
PY
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_FLAT SQ_INSTS_SMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
cat $O/x3_train_gemm_pmc.txt | cut -c1-260
