#!/bin/bash
# A/B of the host-feed upload paths in ONE session on one box (Session.upload_many): the bench line's api_rates (numpy in / numpy
# out per step, the stream idle at every D-step feed) and eval rollout, three rounds alternating.  OUT = output file.
OUT=$1
for round in 1 2 3; do
  for mode in pageable staged auto; do
    for fl in "" "--dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10"; do
      ACG_UPLOAD=$mode python3 bench.py --no-cpu-baseline $fl 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $round  %-8s %-55s value %7.1f  plain %7.1f  numpy %7.1f  rollout frames/s %8.1f' % ('$mode', '$fl' or '(config 2)', d['value'], d['api_rates']['plain_call_path_device_resident'], d['api_rates']['numpy_in_numpy_out_as_train_py'], d['eval_rollout']['frames_per_s']))" >> $OUT
    done
  done
done
