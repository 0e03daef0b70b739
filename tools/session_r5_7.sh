#!/bin/bash
# GPU session 7: new GPU tests; soak runs of train() (look-ahead, paired D steps) in f32 / bf16 / wass.
OUT=gpurun_out/s7; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -k "lookahead or tfrecords or pairs" > $OUT/tests_new.log 2>&1; echo "new tests rc $?" | tee -a $OUT/tests_new.log
tail -5 $OUT/tests_new.log
SOAK_ITERS=2000 python tools/soak_train.py f32 bf16 > $OUT/soak.txt 2>&1
SOAK_ITERS=500 python tools/soak_train.py wass wass-bf16 >> $OUT/soak.txt 2>&1
grep -v amdgpu.ids $OUT/soak.txt | tail -8
