#!/bin/bash
# GPU session 5: full GPU suite; bench A/B of the copy-instead-of-DNA-relaunch; train loop with span-reading process workers.
OUT=gpurun_out/s5; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc $?" | tee -a $OUT/tests.log
tail -8 $OUT/tests.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python bench.py --no-cpu-baseline --dtype bf16 > $OUT/bench_bf16.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 > $OUT/bench_c5.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s5/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'steps/s', d['ms_per_step'], 'ms', d.get('api_rates'), {k:v for k,v in d['op_ms_per_step'].items() if 'Dna' in k or 'Copy' in k})
    except Exception as e:
        print(f, 'FAILED', e)
PY
python tools/bench_train_loop.py > $OUT/train_loop.txt 2>&1; grep -v amdgpu.ids $OUT/train_loop.txt
