#!/bin/bash
# Round-6 profile collection on the GPU box. Results -> gpurun_out/r6/prof ; scratch/publish_profiles_r6.py copies the summaries into
# profiles/r6_*. Sections (argv: any of `line cfg1 cfg4 train msda`, default all): the default bench line; rocprofv3 kernel stats + the
# eager per-step table + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ / L2 / GRBM) for configs[1] and configs[4]; per-step kernel
# tables of the training steps (configs[2], configs[3], both precisions); counters of the MSDeformAttn backward kernels.
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r6/prof
mkdir -p $O
WHAT="${@:-line cfg1 cfg4 train msda}"
LEAN="--no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --extra-workloads 0 --no-einsum-sweep"
infer() {   # $1 = workload tag, $2 = delimiter kernel of the eager step table
  W=$1
  BENCH="$R/bench.py --workload $W --steps 20 --warmup 5 $LEAN --repeats 3"
  rm -rf /tmp/stats_$W /tmp/stepprof_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$W -- python3 $BENCH > $O/${W}_bench_under_rocprof.log 2>&1
  cp $(find /tmp/stats_$W -name "*kernel_stats.csv" | head -1) $O/${W}_kernel_stats.csv
  python3 $R/scratch/r4_launches.py /tmp/stats_$W $O/${W}_hot_kernel_launches.json > $O/${W}_hot_kernel_launches.txt
  EAGER="$R/bench.py --workload $W --graph 0 --pipeline 0 --steps 4 --warmup 3 --repeats 1 $LEAN"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/stepprof_$W -- python3 $R/bench.py --workload $W --graph 0 --pipeline 0 --steps 10 --warmup 3 --repeats 1 $LEAN > /dev/null 2>&1
  python3 $R/scratch/step_kernels2.py /tmp/stepprof_$W "$2" 8 > $O/${W}_step_kernels.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${W}_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${W}_$c -- python3 $EAGER > /dev/null 2>&1
    cp $(find /tmp/pmc_${W}_$c -name "*counter_collection.csv" | head -1) $O/${W}_pmc_$c.csv
  done
  rm -rf /tmp/pmc_${W}_sq
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_${W}_sq -- python3 $EAGER > /dev/null 2>&1
  cp $(find /tmp/pmc_${W}_sq -name "*counter_collection.csv" | head -1) $O/${W}_pmc_sq.csv
  rm -rf /tmp/pmc_${W}_grbm
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_${W}_grbm -- python3 $EAGER > /dev/null 2>&1
  cp $(find /tmp/pmc_${W}_grbm -name "*counter_collection.csv" | head -1) $O/${W}_pmc_grbm.csv
}
for what in $WHAT; do
case $what in
line)
  cd $R
  python bench.py > $O/bench_line.json 2> $O/bench_line.err
  cp $R/gpurun_out/bench_full.json $O/bench_full.json
  tail -c 600 $O/bench_line.json
  cd /tmp ;;
cfg1) infer cfg1 cgg_class_topk ;;
cfg4) infer cfg4 cgg_panoptic_paint ;;
train)
  for w in cfg2 cfg3; do for p in fp32 bf16; do
    rm -rf /tmp/trainprof_${w}_$p
    rocprofv3 --kernel-trace --output-format csv -d /tmp/trainprof_${w}_$p -- python3 $R/bench.py --workload $w --steps 4 --warmup 2 --precision $p > $O/${w}_train_under_rocprof_$p.log 2>&1
    # (bench.py runs 2 warm-up + 4 timed + 1 event-timed step = 7 steps; the clip's NormTwoOps launches per step depend on the bucket count)
    python3 $R/scratch/step_kernels2.py /tmp/trainprof_${w}_$p NormTwoOps 3 auto:7 > $O/${w}_train_step_kernels_$p.txt 2>&1
  done; done
  head -30 $O/cfg2_train_step_kernels_fp32.txt | cut -c1-150 ;;
msda)
  M=$O/msda_bwd
  mkdir -p $M
  : > $M/timing.txt
  for std in 0.5 2.0 8.0; do python3 $R/scratch/msda_bwd_only.py $std 5 2>&1 | grep -v amdgpu >> $M/timing.txt; done
  rm -rf /tmp/mb_kt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mb_kt -- python3 $R/scratch/msda_bwd_only.py 2.0 5 > /dev/null 2>&1
  cp $(find /tmp/mb_kt -name "*kernel_stats.csv" | head -1) $M/kernel_stats.csv
  pass() { n=$1; shift
    rm -rf /tmp/mb_$n
    rocprofv3 --pmc "$@" --output-format csv -d /tmp/mb_$n -- python3 $R/scratch/msda_bwd_only.py 2.0 3 > /dev/null 2>&1
    python3 $R/scratch/pmc_filter.py $(find /tmp/mb_$n -name "*counter_collection.csv" | head -1) msda_bwd > $M/pmc_$n.csv
  }
  pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
  pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
  pass grbm GRBM_GUI_ACTIVE
  cat $M/timing.txt ;;
esac
done
ls -la $O | head -50
