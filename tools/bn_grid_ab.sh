#!/bin/bash
# What the two-launch BatchNorm path (taken with more than one rank under side-stream collectives: Session._bn_flags) costs on ONE rank,
# one session, alternating: the default step and the same step with the data-parallel machinery forced on (one-rank RCCL, side stream).
OUT=$1
for round in 1 2; do
  for fl in "" "--force-dp --dp-collectives side" "--dtype bf16" "--dtype bf16 --force-dp --dp-collectives side"; do
    for mode in on not-beside-collectives off; do
      python3 bench.py --no-cpu-baseline --no-api-rates --bn-grid-exchange $mode $fl 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $round  grid exchange %-22s %-50s %7.1f steps/s  %.4f ms/step  (%s)' % ('$mode', '$fl' or '(config 2)', d['value'], d['ms_per_step'], d['config']['bn_grid_exchange']))" >> $OUT
    done
  done
done
