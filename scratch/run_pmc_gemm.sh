#!/bin/bash
# fullsize tests verbosely (a crash was seen after them in the whole-suite run) + counters of the training contractions
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_fullsize_gpu.py -v -m gpu -x -k test_configs3_detector_train_step_runs > gpurun_out/r5/pytest_fullsize.txt 2>&1
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/r5/pytest_fullsize.txt | tail -12
grep -n "Fatal\|Segmentation\|Abort\|Current thread\|Memory access fault" -A14 gpurun_out/r5/pytest_fullsize.txt | head -50
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift
  rm -rf /tmp/xg_$n
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/xg_$n -- python3 $R/scratch/x3_train_gemm_bench.py 344064 3 > /dev/null 2>&1
  python3 $R/scratch/pmc_summary.py $(find /tmp/xg_$n -name "*counter_collection.csv" | head -1) _x3_kernel > $R/gpurun_out/r5/xg_pmc_$n.csv
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass grbm GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
cat $R/gpurun_out/r5/xg_pmc_sq.csv $R/gpurun_out/r5/xg_pmc_sq2.csv $R/gpurun_out/r5/xg_pmc_fetch.csv | cut -c1-160
cat $R/gpurun_out/r5/xg_pmc_grbm.csv | cut -c1-200
