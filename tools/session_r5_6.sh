#!/bin/bash
# GPU session 6: full GPU suite; config-4 bench with paired D steps; numpy call path after the single upload; train loop.
OUT=gpurun_out/s6; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc $?" | tee -a $OUT/tests.log
tail -6 $OUT/tests.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python bench.py --no-cpu-baseline --loss wass --opt rmsprop > $OUT/bench_c4.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --loss wass --opt rmsprop --no-lookahead --no-api-rates > $OUT/bench_c4_nola.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --loss wass --opt rmsprop --dtype bf16 > $OUT/bench_c4_bf16.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s6/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'steps/s', d['ms_per_step'], 'ms', d.get('api_rates'), 'conv', d['roofline']['hot_relaunch'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
python tools/bench_train_loop.py --iters 200 > $OUT/train_loop.txt 2>&1; grep -v amdgpu.ids $OUT/train_loop.txt
