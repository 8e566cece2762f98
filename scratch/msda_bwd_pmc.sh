#!/bin/bash
# round 4: kernel trace + PMC passes (separate, as the guide prescribes) of the tiled MSDeformAttn backward at configs[2] shapes
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r4/msda_bwd
mkdir -p $O
python3 $R/scratch/msda_bwd_only.py 2.0 5 2>&1 | grep -v amdgpu > $O/timing.txt
python3 $R/scratch/msda_bwd_only.py 0.5 5 2>&1 | grep -v amdgpu >> $O/timing.txt
python3 $R/scratch/msda_bwd_only.py 8.0 5 2>&1 | grep -v amdgpu >> $O/timing.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mb_kt -- python3 $R/scratch/msda_bwd_only.py 2.0 5 > /dev/null 2>&1
cp $(find /tmp/mb_kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
pass() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/mb_$n -- python3 $R/scratch/msda_bwd_only.py 2.0 3 > /dev/null 2>&1
  python3 $R/scratch/pmc_filter.py $(find /tmp/mb_$n -name "*counter_collection.csv" | head -1) msda_bwd_tiled > $O/pmc_$n.csv
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
pass grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/r4/msda_bwd/pmc_*.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k: round(sum(v) / len(v)) for k, v in agg.items()}, 'launches', max(len(v) for v in agg.values()) if agg else 0)
PY
cat $O/timing.txt; head -5 $O/kernel_stats.csv | cut -c1-200
