#!/bin/bash
# What the convolution kernels do with their cycles over REAL steps: one rocprofv3 --pmc pass (8 SQ slots + GRBM) of the bench's own
# replayed steps, reduced per kernel by tools/pmc_mfma_summarise.py.   usage: bash tools/pmc_mfma.sh OUTDIR [bench.py flags]
OUT=${1:-gpurun_out/pmc_mfma}; shift
mkdir -p $OUT && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/sq -- python3 bench.py --trace-run --steps 4 --warmup 1 --min-seconds 0 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1
PYTHONPATH=tools python3 tools/pmc_mfma_summarise.py $OUT/sq
