O=gpurun_out/s7; mkdir -p $O
L=action_conditioned_gans_amd/csrc/libacgan_hip_convtune.so
echo "== wide (256x128 LDS-DMA), M = 65536 = 256 tiles"; python3 tools/conv16_probe.py --batch 64 --cins 64,128,256,512 --which fwd --lib $L 2>/dev/null
echo "== 128x128 register-staged"; ACG_PLAN16_WIDE_TILES=1000000 python3 tools/conv16_probe.py --batch 64 --cins 64,128,256,512 --which fwd --lib $L 2>/dev/null
echo "== wide, M = 131072 = 512 tiles"; python3 tools/conv16_probe.py --batch 128 --cins 64,128,256 --which fwd --lib $L 2>/dev/null
echo "== 128x128, M = 131072"; ACG_PLAN16_WIDE_TILES=1000000 python3 tools/conv16_probe.py --batch 128 --cins 64,128,256 --which fwd --lib $L 2>/dev/null
