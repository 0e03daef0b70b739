O=gpurun_out/s10; mkdir -p $O
L=action_conditioned_gans_amd/csrc/libacgan_hip_convtune.so

C5="--dtype bf16 --img 128 --ksize 11"
ACG_CONV16_KORDER=0 python3 tools/conv_table.py $C5 --lib $L > $O/conv_c5_k0.txt 2>/dev/null
ACG_CONV16_KORDER=1 python3 tools/conv_table.py $C5 --lib $L > $O/conv_c5_k1.txt 2>/dev/null
ACG_CONV16_KORDER=0 python3 tools/conv_table.py --dtype bf16 --lib $L > $O/conv_c3_k0.txt 2>/dev/null
ACG_CONV16_KORDER=1 python3 tools/conv_table.py --dtype bf16 --lib $L > $O/conv_c3_k1.txt 2>/dev/null
head -1 $O/conv_*.txt
B5="--no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10"
for r in 1 2; do for v in 0 1; do ACG_CONV16_KORDER=$v python3 bench.py $B5 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 korder $v', d['value'], d['ms_per_step'], d['roofline']['hot_relaunch']['achieved'])"; done; done
for r in 1 2; do for v in 0 1; do ACG_CONV16_KORDER=$v python3 bench.py --no-cpu-baseline --dtype bf16 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 korder $v', d['value'], d['ms_per_step'], d['roofline']['hot_relaunch']['achieved'])"; done; done
