#!/bin/bash
# The agreement records bench.py publishes (profiles/r6_agreement.json): measured by tests/test_fullsize_gpu.py at the BASELINE
# configs' real shapes against the f32 / float64 CPU oracle.
#   on the GPU box:   gpurun --timeout 2400 -- 'bash tools/collect_agreement.sh run'   -> gpurun_out/fullsize_agreement.json
#   back here:        bash tools/collect_agreement.sh publish                          -> profiles/r6_agreement.json (all keys checked)
set -e
cd "$(dirname "$0")/.."
KEYS="configs1_fp32_no_injection configs1_bf16 configs3_bf16 configs4_bf16 configs2_train_slice configs3_train_slice"
case "${1:-run}" in
  run)
    rm -f gpurun_out/fullsize_agreement.json
    python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x 2>&1 | tail -15
    ;;
  publish)
    python - $KEYS <<'PY'
import json, sys
d = json.load(open('gpurun_out/fullsize_agreement.json'))
missing = [k for k in sys.argv[1:] if not isinstance(d.get(k), dict)]
if missing:
    raise SystemExit(f'gpurun_out/fullsize_agreement.json lacks {missing}: run the WHOLE of tests/test_fullsize_gpu.py (tools/collect_agreement.sh run)')
json.dump(d, open('profiles/r6_agreement.json', 'w'), indent=1)
print('profiles/r6_agreement.json:', sorted(d))
PY
    ;;
esac
