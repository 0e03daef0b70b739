#!/bin/bash
# A/B inside ONE box: a layer's two gradients as one launch (default) vs two launches (--no-pair).
set -e
OUT=$1; mkdir -p $OUT
for rep in 1 2; do
for f in "" "--no-pair"; do
  python3 bench.py --dtype bf16 --no-cpu-baseline $f > $OUT/c3_${rep}_${f#--}.json 2>$OUT/err.txt
  python3 bench.py --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --no-cpu-baseline $f > $OUT/c5_${rep}_${f#--}.json 2>>$OUT/err.txt
done; done
python3 - <<PY
import json,glob
for p in sorted(glob.glob('$OUT/c*.json')):
    j=json.loads(open(p).read().strip().splitlines()[-1]); print(p.split('/')[-1], j['value'], j['ms_per_step'], j['roofline']['achieved'])
PY
