O=gpurun_out/s12; mkdir -p $O
L=action_conditioned_gans_amd/csrc/libacgan_hip_bntune.so
for r in 1 2; do
python3 bench.py --no-cpu-baseline --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 fused', d['value'], d['ms_per_step'], d['op_ms_per_step'].get('BnActBwdOp'))"
ACG_BN_FUSED_BWD=0 python3 bench.py --no-cpu-baseline --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 unfused', d['value'], d['ms_per_step'], d['op_ms_per_step'].get('BnActBwdOp'))"
done
python3 bench.py --no-cpu-baseline --dtype bf16 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 fused', d['value'], d['ms_per_step'])"
ACG_BN_FUSED_BWD=0 python3 bench.py --no-cpu-baseline --dtype bf16 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 unfused', d['value'], d['ms_per_step'])"
B5="--no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10"
python3 bench.py $B5 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 fused', d['value'], d['ms_per_step'])"
ACG_BN_FUSED_BWD=0 python3 bench.py $B5 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 unfused', d['value'], d['ms_per_step'])"
python3 bench.py --no-cpu-baseline --loss wass --opt rmsprop --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 fused', d['value'], d['ms_per_step'])"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -6 $O/gputests.log
