#!/bin/bash
# GPU session 3 of round 5: full GPU suite after the feed-alias fix, train-loop rates with process workers, bench A/B.
OUT=gpurun_out/s3; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc $?" | tee -a $OUT/tests.log
tail -12 $OUT/tests.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python bench.py --no-cpu-baseline --dtype bf16 > $OUT/bench_bf16.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --loss wass --opt rmsprop > $OUT/bench_c4.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --loss wass --opt rmsprop --no-lookahead --no-api-rates > $OUT/bench_c4_nola.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 > $OUT/bench_c5.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --no-lookahead --no-api-rates > $OUT/bench_c5_nola.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s3/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'steps/s', d['ms_per_step'], 'ms', d.get('api_rates'), 'dna', (d.get('roofline_dna') or {}).get('hot_relaunch'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
python tools/bench_train_loop.py > $OUT/train_loop.txt 2>&1; grep -v amdgpu.ids $OUT/train_loop.txt
for lib in action_conditioned_gans_amd/csrc/libacgan_hip.so build_tools/libacgan_dna_lane_ty1.so build_tools/libacgan_dna_lane_ty2.so; do
  for dt in f32 bf16; do
    echo "== $lib $dt" >> $OUT/dna_k11_ab.txt
    python tools/bench_dna.py --lib $lib --img 128 --ksize 11 --dtype $dt --batches 8,32 >> $OUT/dna_k11_ab.txt 2>&1
  done
done
grep -v amdgpu.ids $OUT/dna_k11_ab.txt
