"""bench.py's final stdout line must stay driver-readable (round 5's 36-KB line came back `parsed: null`): the compact emitter is
run here on the full record of a real run (profiles/r5_bench_line.json, 36 KB) and on hostile synthetic records."""
import io
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _strict(line):
    def bad(x):
        raise ValueError(f'non-JSON constant {x}')
    return json.loads(line, parse_constant=bad)


def _emit(rec, tmp_path):
    buf = io.StringIO()
    line = bench.emit(rec, full_path=str(tmp_path / 'full.json'), stream=buf)
    out = buf.getvalue()
    assert out.endswith('\n') and out.count('\n') == 1 and out.strip() == line
    assert len(line.encode()) < 4096 <= bench.LINE_LIMIT
    return _strict(line), _strict(open(tmp_path / 'full.json').read())


def test_real_record_compacts_under_4k_with_every_contract_key(tmp_path):
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r5_bench_line.json')))
    assert len(json.dumps(full)) > 30000                      # the record that broke the driver's parser
    d, kept = _emit(full, tmp_path)
    for k in CONTRACT:
        assert k in d, k
    assert d['value'] == float(f'{full["value"]:.6g}') and d['unit'] == 'images/sec' and d['n_gpus'] == 1
    assert d['steps'] == full['steps'] and d['warmup'] == full['warmup'] and d['vs_baseline'] is None
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert set(d['config']) >= {'workload', 'global_batch', 'parallelism'} and 'model' not in d['config']
    assert len(d['config']['workload']) <= 120
    r = d['roofline']
    assert r['kernel'] == 'cgg_gemm_x3s_kernel' and r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s'
    assert math.isclose(r['frac'], r['achieved'] / r['peak'], rel_tol=2e-3)
    assert math.isclose(r['frac'], full['roofline']['frac'], rel_tol=1e-3) and r['traffic'] == float(f'{full["roofline"]["traffic"]:.4g}')
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == full['cpu_baseline']['cores'] and c['unit'] == 'images/sec' and len(c['sample']) <= 100
    assert math.isclose(c['value'], full['cpu_baseline']['value'], rel_tol=1e-3)
    assert math.isclose(d['train_step']['value'], full['train_step']['value'], rel_tol=1e-3)
    assert math.isclose(d['train_step']['bf16']['value'], full['train_step']['bf16_mode']['value'], rel_tol=1e-3)
    assert math.isclose(d['extra']['cfg3']['value'], full['extra']['configs[3]']['value'], rel_tol=1e-3)
    assert math.isclose(d['extra']['cfg4']['value'], full['extra']['configs[4]']['value'], rel_tol=1e-3)
    assert d['einsum']['best']['frac_bf16_mfma_peak'] > 0.1 and len(d['einsum']['rows']) == 10
    assert kept['value'] == full['value'] and kept['roofline'] == full['roofline']      # the full record is kept verbatim beside it
    assert 'dropped_for_length' not in d


def test_hostile_record_still_one_strict_line(tmp_path):
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r5_bench_line.json')))
    full['config']['workload'] = 'w' * 5000
    full['roofline']['kernel'] = 'cgg_kernel<' + 'T, ' * 2000 + '> (prose)'
    full['roofline']['achieved'] = float('nan')
    full['cpu_baseline']['sample'] = 's' * 9000
    full['einsum_mfma_target'] = [dict(r, batch=2, form='f' * 30) for r in full['einsum_mfma_target'] if 'queries' in r] * 40
    full['train_step'] = dict(error='child exited 1' + 'x' * 5000)
    full['latency_ms_per_batch'] = float('inf')
    d, _ = _emit(full, tmp_path)
    for k in CONTRACT:
        assert k in d, k
    assert d['roofline']['achieved'] is None and d['roofline']['kernel'] == 'cgg_kernel'
    assert 'einsum' in d.get('dropped_for_length', [])        # the optional objects go first, never roofline / cpu_baseline
    assert d.get('latency_ms_per_batch') is None


def test_training_workload_line(tmp_path):
    ts = json.load(open(os.path.join(ROOT, 'profiles', 'r5_bench_line.json')))['train_step']
    rec = dict(metric='images/sec (training step, 1024x1024, 100 queries)', value=ts['value'], unit='images/sec', n_gpus=1, steps=5, warmup=3,
               ms_per_step=ts['ms_per_step'], higher_is_better=True, scaling='weak', vs_baseline=None, dtype=ts['dtype'], data='synthetic',
               config=dict(workload=ts['workload'], global_batch=16, parallelism='dp1', precision='fp32'), loss=ts['loss'],
               roofline=ts['roofline'], kernels=ts['kernels'])
    d, _ = _emit(rec, tmp_path)
    assert d['dtype'] == 'f32' and d['roofline']['frac'] > 0 and d['config']['parallelism'] == 'dp1' and 'loss' in d


def test_committed_agreement_profile_has_every_record_the_line_publishes():
    recs, src = bench.agreement_records()
    missing = [k for k in bench.AGREEMENT_KEYS if not isinstance(recs.get(k), dict)]
    assert not missing, f'{src} lacks {missing}: tools/collect_agreement.sh'
