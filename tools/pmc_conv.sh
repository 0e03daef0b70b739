# rocprofv3 PMC passes over representative conv launches (usage: bash tools/pmc_conv.sh OUTDIR)
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT && export TMPDIR=/tmp
for spec in "d/conv2/conv2d 3 2" "g/tconv3/conv2d_transpose 3 1" "g/conv4/conv2d 3 8" "g/tconv3/conv2d_transpose/wgrad 0 4" "d/conv2/conv2d/wgrad 1 8"; do
  set -- $spec
  tag=$(echo $1 | tr '/' '_')_c$2_s$3
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/a_$tag -- python3 tools/tune_conv.py --only $1 --cfgs $2 --splits $3 --reps 5 > $OUT/a_$tag.log 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/b_$tag -- python3 tools/tune_conv.py --only $1 --cfgs $2 --splits $3 --reps 5 > $OUT/b_$tag.log 2>&1
done
ls $OUT | head -30
