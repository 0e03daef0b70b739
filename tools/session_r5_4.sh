#!/bin/bash
OUT=gpurun_out/s4; mkdir -p $OUT
python tools/debug/lookahead_divergence.py > $OUT/lookahead_divergence.txt 2>&1; grep -v amdgpu.ids $OUT/lookahead_divergence.txt
