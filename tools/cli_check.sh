#!/bin/bash
# End-to-end check of the CLI on a GPU box: writes 64 synthetic push records, runs `python -m action_conditioned_gans_amd.train` on them with
# the CLI defaults (process decode workers, announced frames, 8 GiB frame cache, background checkpoints), lists what it left behind.
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import bench_train_loop as BL
os.makedirs('/tmp/push_cli', exist_ok=True)
BL.make_shards('/tmp/push_cli', 64)
print('shards ok')
PY
rm -rf /tmp/cli_out
time python -m action_conditioned_gans_amd.train /tmp/push_cli /tmp/cli_out --adv True --dna --batch_size 32 --train_iter 450 > /tmp/cli_stdout.txt 2> /tmp/cli_stderr.txt
echo "exit $?"
tail -3 /tmp/cli_stdout.txt
ls /tmp/cli_out /tmp/cli_out/models | head -20
wc -l /tmp/cli_out/logs/train.jsonl
tail -2 /tmp/cli_stderr.txt
