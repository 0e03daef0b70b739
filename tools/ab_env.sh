#!/bin/bash
# Sweep of one tuning-build environment knob inside ONE box:  tools/ab_env.sh OUTDIR VAR v1 v2 ...   (three workloads each)
set -e
OUT=$1; VAR=$2; shift 2; mkdir -p $OUT
L=action_conditioned_gans_amd/csrc/libacgan_hip_tuning.so
for rep in 1 2; do
for v in "$@"; do
  env $VAR=$v python3 bench.py --lib $L --no-cpu-baseline > $OUT/f32_${v}_$rep.json 2>$OUT/err.txt
  env $VAR=$v python3 bench.py --lib $L --dtype bf16 --no-cpu-baseline > $OUT/c3_${v}_$rep.json 2>>$OUT/err.txt
  env $VAR=$v python3 bench.py --lib $L --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --no-cpu-baseline > $OUT/c5_${v}_$rep.json 2>>$OUT/err.txt
done; done
python3 - <<PY
import json,glob
for p in sorted(glob.glob('$OUT/*.json')):
    j=json.loads(open(p).read().strip().splitlines()[-1]); print('%-24s steps/s %8.2f  ms/step %.4f' % (p.split('/')[-1][:-5], j['value'], j['ms_per_step']))
PY
