#!/bin/bash
# A/B of one bench.py flag inside ONE box, two repeats, the three workloads:  tools/ab_flag.sh OUTDIR --some-flag
set -e
OUT=$1; FLAG=$2; mkdir -p $OUT
for rep in 1 2; do
for f in "" "$FLAG"; do
  python3 bench.py --no-cpu-baseline $f > $OUT/f32_${rep}_${f#--}.json 2>$OUT/err.txt
  python3 bench.py --dtype bf16 --no-cpu-baseline $f > $OUT/c3_${rep}_${f#--}.json 2>>$OUT/err.txt
  python3 bench.py --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --no-cpu-baseline $f > $OUT/c5_${rep}_${f#--}.json 2>>$OUT/err.txt
done; done
python3 - <<PY
import json,glob
for p in sorted(glob.glob('$OUT/*.json')):
    j=json.loads(open(p).read().strip().splitlines()[-1]); print('%-30s steps/s %8.2f  ms/step %.4f  conv TF/s %.1f' % (p.split('/')[-1][:-5], j['value'], j['ms_per_step'], j['roofline']['achieved']))
PY
