#!/bin/bash
# GPU session 8: BatchNorm fused-kernel changes (local totals for one row block, wide blocks for slab inputs): tests + A/B against the previous library.
OUT=gpurun_out/s8; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc $?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
for rep in 1 2; do
for lib in build_tools/libacgan_prev.so action_conditioned_gans_amd/csrc/libacgan_hip.so; do
  for flags in "" "--dtype bf16"; do
    python bench.py --no-cpu-baseline --no-api-rates --lib $lib $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$flags', d['value'], 'steps/s', d['ms_per_step'], {k:v for k,v in d['op_ms_per_step'].items() if 'Bn' in k})" | tee -a $OUT/bn_wide_ab.txt
  done
done
done
python tools/conv_table.py --other > $OUT/other_new.txt 2>/dev/null; python tools/conv_table.py --other --lib build_tools/libacgan_prev.so > $OUT/other_prev.txt 2>/dev/null
grep BnAct $OUT/other_new.txt | sort > /tmp/n.txt; grep BnAct $OUT/other_prev.txt | sort > /tmp/p.txt; join -j 2 <(awk '{print $1"_"$2, $NF, $(NF-1)}' /tmp/p.txt | sort) <(awk '{print $1"_"$2, $NF, $(NF-1)}' /tmp/n.txt | sort) 2>/dev/null | head -5
paste <(awk '{printf "%s %s %s ", $1, $2, $4; print $(NF-1)}' /tmp/p.txt) <(awk '{print $(NF-1)}' /tmp/n.txt) | awk '{d=$4-$5; printf "%-4s %-40s %-18s prev %6.1f new %6.1f  %+5.1f\n", $1,$2,$3,$4,$5,-d}' | sort -k8 -n | tee $OUT/bn_per_op_ab.txt | head -60
