#!/bin/bash
# counters of the mask-logit einsum kernels (scratch/einsum_only.py), separate rocprofv3 --pmc passes -> gpurun_out/r6/einsum_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r6
mkdir -p $O
: > $O/einsum_pmc.txt
pass() { n=$1; shift
  rm -rf /tmp/ep_$n
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/ep_$n -- python3 $R/scratch/einsum_only.py > /dev/null 2>&1
  python3 $R/scratch/pmc_filter.py $(find /tmp/ep_$n -name "*counter_collection.csv" | head -1) mask_logits > /tmp/ep_$n.csv
  python3 - /tmp/ep_$n.csv $n >> $O/einsum_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r['Kernel_Name'].split('(')[0][:70], r.get('Grid_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), d in sorted(agg.items()):
    print('%s  %s grid %s: ' % (sys.argv[2], k, g) + ', '.join('%s %.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())) + '  (%d launches)' % max(len(v) for v in d.values()))
PY
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
cat $O/einsum_pmc.txt | cut -c1-260
