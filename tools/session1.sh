set -x
mkdir -p gpurun_out/s1
python3 tools/bench_bn.py --check > gpurun_out/s1/bn_new_f32_c2.txt 2>&1 || exit 1
python3 tools/bench_bn.py --lib action_conditioned_gans_amd/csrc/libacgan_hip_bn_r3.so > gpurun_out/s1/bn_r3_f32_c2.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --check > gpurun_out/s1/bn_new_bf16_c2.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --lib action_conditioned_gans_amd/csrc/libacgan_hip_bn_r3.so > gpurun_out/s1/bn_r3_bf16_c2.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --set c5 > gpurun_out/s1/bn_new_bf16_c5.txt 2>&1
python3 tools/bench_bn.py --dtype bf16 --set c5 --lib action_conditioned_gans_amd/csrc/libacgan_hip_bn_r3.so > gpurun_out/s1/bn_r3_bf16_c5.txt 2>&1
python3 tools/bench_bn.py --rot 10 > gpurun_out/s1/bn_new_f32_c2_rot10.txt 2>&1
python3 tools/bench_bn.py --rot 10 --lib action_conditioned_gans_amd/csrc/libacgan_hip_bn_r3.so > gpurun_out/s1/bn_r3_f32_c2_rot10.txt 2>&1
tail -n 12 gpurun_out/s1/bn_*_f32_c2.txt
python3 bench.py --no-cpu-baseline > gpurun_out/s1/bench_new.json 2>gpurun_out/s1/bench_new.err
python3 bench.py --no-cpu-baseline --lib action_conditioned_gans_amd/csrc/libacgan_hip_bn_r3.so > gpurun_out/s1/bench_r3bn.json 2>gpurun_out/s1/bench_r3bn.err
python3 bench.py --no-cpu-baseline --dtype bf16 > gpurun_out/s1/bench_new_bf16.json 2>>gpurun_out/s1/bench_new.err
python3 bench.py --no-cpu-baseline --dtype bf16 --lib action_conditioned_gans_amd/csrc/libacgan_hip_bn_r3.so > gpurun_out/s1/bench_r3bn_bf16.json 2>>gpurun_out/s1/bench_r3bn.err
python3 -c "
import json
for f in ('bench_new','bench_r3bn','bench_new_bf16','bench_r3bn_bf16'):
    d=json.loads(open('gpurun_out/s1/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['hot_relaunch']['frac'], d['op_ms_per_step'].get('BnActOp'), d['op_ms_per_step'].get('BnActBwdOp'))
"
bash tools/pmc_bn.sh gpurun_out/s1/pmc_bn_f32_c2 > /dev/null 2>&1
bash tools/pmc_bn.sh gpurun_out/s1/pmc_bn_bf16_c5 --dtype bf16 --set c5 > /dev/null 2>&1
head -50 gpurun_out/s1/pmc_bn_f32_c2/summary.txt
