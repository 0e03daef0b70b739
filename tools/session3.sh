set -x
O=gpurun_out/s3; mkdir -p $O
L=action_conditioned_gans_amd/csrc
python3 tools/bench_bn.py --check > $O/bn_new_f32_c2.txt 2>&1 || { tail -30 $O/bn_new_f32_c2.txt; exit 1; }
python3 tools/bench_bn.py --lib $L/libacgan_hip_bn_r3.so > $O/bn_r3_f32_c2.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --check > $O/bn_new_bf16_c2.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --lib $L/libacgan_hip_bn_r3.so > $O/bn_r3_bf16_c2.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --set c5 --check > $O/bn_new_bf16_c5.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --set c5 --lib $L/libacgan_hip_bn_r3.so > $O/bn_r3_bf16_c5.txt 2>&1
for nb in 256 1024 2048; do
  ACG_BN_APPLY_BLOCKS=$nb python3 tools/bench_bn.py --lib $L/libacgan_hip_bntune.so > $O/bn_tune_f32_c2_blocks$nb.txt 2>&1
  ACG_BN_APPLY_BLOCKS=$nb python3 tools/bench_bn.py --dtype bf16 --set c5 --lib $L/libacgan_hip_bntune.so > $O/bn_tune_bf16_c5_blocks$nb.txt 2>&1
done
ACG_BN_CL16=1 python3 tools/bench_bn.py --dtype bf16 --lib $L/libacgan_hip_bntune.so > $O/bn_tune_bf16_c2_cl16.txt 2>&1
ACG_BN_CL16=1 python3 tools/bench_bn.py --dtype bf16 --set c5 --lib $L/libacgan_hip_bntune.so > $O/bn_tune_bf16_c5_cl16.txt 2>&1
ACG_BN_FINALIZE_BLOCKS=100000 python3 tools/bench_bn.py --dtype bf16 --set c5 --lib $L/libacgan_hip_bntune.so > $O/bn_tune_bf16_c5_nofinalize.txt 2>&1
ACG_BN_FINALIZE_BLOCKS=255 python3 tools/bench_bn.py --lib $L/libacgan_hip_bntune.so > $O/bn_tune_f32_c2_finalize255.txt 2>&1
grep -h "total" $O/bn_*.txt /dev/null; for f in $O/bn_*.txt; do echo "$f: $(grep total $f)"; done
python3 bench.py --no-cpu-baseline > $O/bench_new.json 2>$O/bench_new.err
python3 bench.py --no-cpu-baseline --lib $L/libacgan_hip_bn_r3.so > $O/bench_r3bn.json 2>$O/bench_r3bn.err
python3 bench.py --no-cpu-baseline --dtype bf16 > $O/bench_new_bf16.json 2>>$O/bench_new.err
python3 bench.py --no-cpu-baseline --dtype bf16 --lib $L/libacgan_hip_bn_r3.so > $O/bench_r3bn_bf16.json 2>>$O/bench_r3bn.err
python3 bench.py --no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 > $O/bench_new_c5.json 2>>$O/bench_new.err
python3 bench.py --no-cpu-baseline --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --lib $L/libacgan_hip_bn_r3.so > $O/bench_r3bn_c5.json 2>>$O/bench_r3bn.err
python3 -c "
import json
for f in ('bench_new','bench_r3bn','bench_new_bf16','bench_r3bn_bf16','bench_new_c5','bench_r3bn_c5'):
    d=json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['hot_relaunch']['frac'], d['op_ms_per_step'].get('BnActOp'), d['op_ms_per_step'].get('BnActBwdOp'))
"
