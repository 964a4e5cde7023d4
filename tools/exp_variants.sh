#!/bin/bash
# scratch experiment runner (not part of the product): time several kernel builds at several batch sizes
for so in "$@"; do
  for u in ${UNITS:-1024 4096 16384}; do
    MSPACK_HIP_SO=$so timeout 200 python bench.py --steps 5 --warmup 2 --exp --units $u 2>&1 | python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so units', d['config']['units_per_gpu'], d['value'], 'MB/s', d['ms_per_step'], 'ms', 'exact' if d['config']['bit_exact'] else 'INEXACT')
except Exception as e: print('$so fail', e)"
  done
done
