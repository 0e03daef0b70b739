O=gpurun_out/s6; mkdir -p $O
L=action_conditioned_gans_amd/csrc/libacgan_hip_convtune.so
C5="--dtype bf16 --img 128 --ksize 11"
ACG_PLAN16_WIDE_TILES=1000000 python3 tools/conv_table.py $C5 --lib $L > $O/conv_c5_off.txt 2>/dev/null
python3 tools/conv_table.py $C5 --lib $L > $O/conv_c5_wide192.txt 2>/dev/null
ACG_PLAN16_WIDE_TILES=128 python3 tools/conv_table.py $C5 --lib $L > $O/conv_c5_wide128.txt 2>/dev/null
ACG_PAIR_WIDE=1 python3 tools/conv_table.py $C5 --lib $L > $O/conv_c5_wide192_pairwide.txt 2>/dev/null
ACG_PLAN16_WIDE_TILES=1000000 python3 tools/conv_table.py --dtype bf16 --lib $L > $O/conv_c3_off.txt 2>/dev/null
ACG_PLAN16_WIDE_TILES=64 python3 tools/conv_table.py --dtype bf16 --lib $L > $O/conv_c3_wide64.txt 2>/dev/null
head -3 $O/conv_*.txt
B5="--no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10"
for v in 1000000 192 128; do ACG_PLAN16_WIDE_TILES=$v python3 bench.py $B5 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 wide_min $v', d['value'], d['ms_per_step'], d['roofline']['hot_relaunch']['achieved'])"; done
ACG_PAIR_WIDE=1 python3 bench.py $B5 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 pairwide', d['value'], d['ms_per_step'], d['roofline']['hot_relaunch']['achieved'])"
for v in 1000000 192; do ACG_PLAN16_WIDE_TILES=$v python3 bench.py --no-cpu-baseline --dtype bf16 --batch 256 --steps 10 --lib $L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('b256 wide_min $v', d['value'], d['ms_per_step'], d['roofline']['hot_relaunch']['achieved'])"; done
