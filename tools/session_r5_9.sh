#!/bin/bash
OUT=gpurun_out/s9; mkdir -p $OUT
python tools/list_wgrad_slabs.py > $OUT/wgrad_slabs.txt 2>&1; grep -v amdgpu $OUT/wgrad_slabs.txt
python -m pytest tests/test_gpu_train.py -m gpu -q -k lookahead 2>&1 | tail -3
