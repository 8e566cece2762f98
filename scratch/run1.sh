python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msda or astat" 2>&1 | tail -25
python -m pytest tests/test_x3s_gpu.py -x -q -m gpu -k "ffn_node or absmax or gradient_magnitudes" 2>&1 | tail -8
python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -8
python scratch/einsum_sweep.py 2>&1 | grep -v amdgpu | cut -c1-260
bash scratch/msda_bwd_r5.sh 2>&1 | grep -v "^\"\|^c=" | tail -14
python bench.py --workload cfg2 --steps 5 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 fp32', d['value'], d['ms_per_step'], d['loss']); print(d['roofline']['family'], d['roofline']['ms_per_step'], {k:(v['ms_per_step']) for k,v in d['kernels'].items()})
"
python bench.py --workload cfg2 --steps 5 --warmup 3 --precision bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 bf16', d['value'], d['ms_per_step'], d['loss'])
"
