O=gpurun_out/s11; mkdir -p $O
L=action_conditioned_gans_amd/csrc/libacgan_hip_bntune.so
timeout -k 10 120 python3 tools/bench_bn.py --check > $O/bn_fused_f32_c2.txt 2>&1 || { tail -20 $O/bn_fused_f32_c2.txt; exit 1; }
grep -v amdgpu $O/bn_fused_f32_c2.txt
ACG_BN_FUSED_BWD=0 timeout -k 10 120 python3 tools/bench_bn.py --lib $L > $O/bn_unfused_f32_c2.txt 2>&1; grep -v amdgpu $O/bn_unfused_f32_c2.txt | tail -11
timeout -k 10 120 python3 tools/bench_bn.py --dtype bf16 --check > $O/bn_fused_bf16_c2.txt 2>&1; grep -v amdgpu $O/bn_fused_bf16_c2.txt | grep -v "check fwd"
timeout -k 10 120 python3 tools/bench_bn.py --dtype bf16 --set c5 --check > $O/bn_fused_bf16_c5.txt 2>&1; grep -v amdgpu $O/bn_fused_bf16_c5.txt | grep -v "check fwd"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bn" > $O/bn_tests.log 2>&1; tail -4 $O/bn_tests.log
