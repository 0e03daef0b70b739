O=gpurun_out/s9; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "opt_step_prepared or optimizers or weights_prepare" > $O/opt_tests.log 2>&1; tail -5 $O/opt_tests.log

python3 bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 fused refresh', d['value'], d['ms_per_step'], d['op_ms_per_step'].get('StepOp'))"
python3 bench.py --no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 fused refresh', d['value'], d['ms_per_step'], d['op_ms_per_step'].get('StepOp'))"
python3 tools/conv_table.py --dtype bf16 --other 2>/dev/null | grep -i "StepOp\|total\|non-conv"
