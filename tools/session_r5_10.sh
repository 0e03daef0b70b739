#!/bin/bash
# GPU session 10: whole-step sweep of the split planner's targets with the look-ahead step (tuning build made on the box).
OUT=gpurun_out/s10; mkdir -p $OUT
( make -s -j16 -C action_conditioned_gans_amd/csrc tuning > $OUT/tuning_build.log 2>&1 && echo tuning built ) || echo tuning build failed
T=action_conditioned_gans_amd/csrc/libacgan_hip_tuning.so
run() { env "$@" python bench.py --no-cpu-baseline --no-api-rates --profile-repeats 1 --lib $T $FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %8.2f steps/s  %.4f ms' % ('$*', d['value'], d['ms_per_step']))" | tee -a $OUT/planner_sweep.txt; }
for FLAGS in "" "--dtype bf16"; do
echo "== bench.py $FLAGS" | tee -a $OUT/planner_sweep.txt
run X=0
run ACG_PLAN_TARGET_FD=128
run ACG_PLAN_TARGET_FD=192
run ACG_PLAN_TARGET_FD=384
run ACG_PLAN_TARGET_FD=512
run ACG_PLAN_TARGET_W=256
run ACG_PLAN_TARGET_W=384
run ACG_PLAN_TARGET_W=768
run ACG_PLAN_MIN_STEPS=2
run ACG_PLAN_MIN_STEPS=6
run ACG_PLAN_MIN_STEPS=8
run ACG_BN_SLAB_WIDE=0
run ACG_BN_FUSED_SMALL_ROWBLOCKS=8
run ACG_BN_FUSED_SMALL_ROWBLOCKS=32
run X=0
done
