#!/bin/bash
# GPU session 1 of round 5: full GPU suite, look-ahead A/B, BatchNorm apply-shape experiments (tuning build made on the box).
OUT=gpurun_out/s1; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
for flags in "" "--no-lookahead" "--dtype bf16" "--dtype bf16 --no-lookahead"; do
  name=$(echo "bench$flags" | tr ' -' '__')
  python bench.py --no-cpu-baseline $flags > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['value'], 'steps/s', d['ms_per_step'], 'ms', 'lookahead', d['config'].get('lookahead'), 'api', d.get('api_rates'), 'conv hot', d['roofline']['hot_relaunch']['achieved'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
( make -s -j16 -C action_conditioned_gans_amd/csrc tuning > $OUT/tuning_build.log 2>&1 && echo tuning built ) || echo tuning build failed
T=action_conditioned_gans_amd/csrc/libacgan_hip_tuning.so
for env in "" "ACG_BN_FINALIZE_BLOCKS=0" "ACG_BN_FINALIZE_BLOCKS=0 ACG_BN_APPLY_BLOCKS=1024" "ACG_BN_FINALIZE_BLOCKS=0 ACG_BN_APPLY_BLOCKS=2048" "ACG_BN_FINALIZE_BLOCKS=0 ACG_BN_APPLY_BLOCKS=4096" "ACG_BN_APPLY_BLOCKS=1024" "ACG_BN_APPLY_BLOCKS=256"; do
  echo "== $env" >> $OUT/bn_apply_ab.txt
  env $env python tools/bench_bn.py --lib $T --set c2 >> $OUT/bn_apply_ab.txt 2>&1
done
echo "== c5 default" >> $OUT/bn_apply_ab.txt; python tools/bench_bn.py --lib $T --set c5 --dtype bf16 >> $OUT/bn_apply_ab.txt 2>&1
echo "== c5 ACG_BN_FINALIZE_BLOCKS=0 ACG_BN_APPLY_BLOCKS=2048" >> $OUT/bn_apply_ab.txt; ACG_BN_FINALIZE_BLOCKS=0 ACG_BN_APPLY_BLOCKS=2048 python tools/bench_bn.py --lib $T --set c5 --dtype bf16 >> $OUT/bn_apply_ab.txt 2>&1
tail -30 $OUT/bn_apply_ab.txt
