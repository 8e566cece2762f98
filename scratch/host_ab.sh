#!/bin/bash
cd /root/repo
for t in 0 0 0 8; do
CGG_RLE_THREADS=$t python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 1 --train-step 0 --no-einsum-sweep --repeats 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads $t:', round(d['host_results']['value'],1), 'img/s; device-only', round(d['value'],1), 'rle bytes/mask', round(d['host_results']['rle_bytes_per_mask']), 'masks/img', d['host_results']['masks_per_image'], '| trained-like', round(d['host_results']['trained_like_masks']['value'],1), round(d['host_results']['trained_like_masks']['rle_bytes_per_mask']), 'threads', d['host_results']['rle_threads'])"
done
nproc
