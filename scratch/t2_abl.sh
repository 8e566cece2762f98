#!/bin/bash
cd /root/repo
for a in 0 1 2 3 4 5 7 8 15 16 32 48 63; do echo -n "ABL=$a: "; CGG_T2_ABL=$a timeout 120 python scratch/tail_x3_bench.py 2>&1 | grep "x3a rows" | cut -c1-120; done
