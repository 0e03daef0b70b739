#!/bin/bash
# GPU session 2 of round 5: the C = 28 BatchNorm failure, the rest of the GPU suite, DNA kernel A/B, optimizer timing, train-loop rates.
OUT=gpurun_out/s2; mkdir -p $OUT
export TMPDIR=/tmp
python tools/debug/bn_c28.py > $OUT/bn_c28.txt 2>&1; tail -25 $OUT/bn_c28.txt
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc $?" | tee -a $OUT/tests.log
tail -15 $OUT/tests.log
for lib in action_conditioned_gans_amd/csrc/libacgan_hip.so build_tools/libacgan_dna_old.so; do
  echo "== $lib" >> $OUT/dna_ab.txt
  python tools/bench_dna.py --lib $lib --batches 32,64,256 >> $OUT/dna_ab.txt 2>&1
done
grep -v amdgpu.ids $OUT/dna_ab.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s2/bench.json').read().strip().splitlines()[-1])
print(d['value'], 'steps/s', d['api_rates'], d['op_ms_per_step'])
PY
python tools/bench_train_loop.py > $OUT/train_loop.txt 2>&1; grep -v amdgpu.ids $OUT/train_loop.txt
