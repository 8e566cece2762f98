python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msda" 2>&1 | tail -4
VARS=0 bash scratch/msda_bwd_r5.sh 2>&1 | tail -3
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mbk -- python3 /root/repo/scratch/msda_bwd_only.py 2.0 5 > /dev/null 2>&1; head -4 $(find /tmp/mbk -name "*kernel_stats.csv" | head -1) | cut -c1-150; cd /root/repo
python bench.py --workload cfg2 --steps 5 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 fp32', d['value'], d['ms_per_step'], d['loss']); print(d['roofline']['family'], d['roofline']['ms_per_step'], {k:(v['ms_per_step']) for k,v in d['kernels'].items()})
"
python bench.py --workload cfg2 --steps 5 --warmup 3 --precision bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 bf16', d['value'], d['ms_per_step'], d['loss'])
"
