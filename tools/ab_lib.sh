#!/bin/bash
# A/B of two library builds inside ONE box (bf16 workloads):  tools/ab_lib.sh OUTDIR path/to/other.so
set -e
OUT=$1; LIB=$2; mkdir -p $OUT
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then L="--lib $LIB"; else L=""; fi
  python3 bench.py --dtype bf16 --no-cpu-baseline $L > $OUT/c3_${rep}_$v.json 2>$OUT/err.txt
  python3 bench.py --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --no-cpu-baseline $L > $OUT/c5_${rep}_$v.json 2>>$OUT/err.txt
done; done
python3 - <<PY
import json,glob
for p in sorted(glob.glob('$OUT/*.json')):
    j=json.loads(open(p).read().strip().splitlines()[-1]); print('%-20s steps/s %8.2f  ms/step %.4f  conv TF/s %.1f' % (p.split('/')[-1][:-5], j['value'], j['ms_per_step'], j['roofline']['achieved']))
PY
