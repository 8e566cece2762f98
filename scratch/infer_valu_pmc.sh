#!/bin/bash
# instruction-mix counters of the configs[1] inference step (eager, one stream): which kernels are VALU-issue bound?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --extra-workloads 0 --no-einsum-sweep"
rm -rf /tmp/ivp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/ivp -- python3 $R/bench.py --workload cfg1 --graph 0 --pipeline 0 --steps 4 --warmup 3 --repeats 1 $LEAN > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('/tmp/ivp/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
rows = []
for k, c in acc.items():
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0          # per-XCD active cycles, summed over the launches
    if cyc <= 0: continue
    rows.append((cyc, k, n[k], 4 * c['SQ_INSTS_VALU'] / (1024 * cyc), c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), c['SQ_INSTS_VALU'] / max(c['SQ_INSTS_MFMA'], 1), c['SQ_INSTS_VALU'] / max(c['SQ_WAVES'], 1), c['SQ_INSTS_SALU'] / max(c['SQ_WAVES'], 1), c['SQ_INSTS_LDS'] / max(c['SQ_WAVES'], 1)))
tot = sum(r[0] for r in rows)
print('share  launches  VALU-busy  MFMA-busy  VALU/MFMA  VALU/wave SALU/wave LDS/wave  kernel')
for cyc, k, nn, vb, mb, vm, vw, sw, lw in sorted(rows, reverse=True)[:28]:
    print(f'{cyc / tot:5.3f}  {nn:5d}    {vb:6.2f}     {mb:6.2f}    {vm:8.1f}  {vw:8.0f}  {sw:7.0f}  {lw:7.0f}  {k}')
PY
