#!/bin/bash
# A/B of library builds inside ONE box, float32 headline:  tools/ab_lib_f32.sh OUTDIR other1.so [other2.so ...]
set -e
OUT=$1; shift; mkdir -p $OUT
for rep in 1 2 3; do
  python3 bench.py --no-cpu-baseline > $OUT/f32_${rep}_new.json 2>$OUT/err.txt
  i=0
  for LIB in "$@"; do i=$((i+1)); python3 bench.py --no-cpu-baseline --lib $LIB > $OUT/f32_${rep}_lib$i.json 2>>$OUT/err.txt; done
done
python3 - <<PY
import json,glob
for p in sorted(glob.glob('$OUT/f32_*.json')):
    j=json.loads(open(p).read().strip().splitlines()[-1]); print('%-20s steps/s %8.2f  ms/step %.4f  conv TF/s %.1f' % (p.split('/')[-1][:-5], j['value'], j['ms_per_step'], j['roofline']['achieved']))
PY
