#!/bin/bash
# kernel times + VALU counters of the MSDeformAttn backward kernels (argv: offset std)
R=$GRAFT_REPO_ROOT
STD=${1:-0.5}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mb_kt /tmp/mb_pmc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mb_kt -- python3 $R/scratch/msda_bwd_only.py $STD 5 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/mb_kt/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'msda' in r['Name'] or 'Fill' in r['Name']: print(r['Name'][:60], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/mb_pmc -- python3 $R/scratch/msda_bwd_only.py $STD 3 > /dev/null 2>&1
python3 $R/scratch/pmc_summary.py $(find /tmp/mb_pmc -name "*counter_collection.csv" | head -1) msda_bwd
