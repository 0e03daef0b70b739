#!/bin/bash
# Round-5 evidence, second half (per-op tables, other configurations) - run after tools/evidence_r5.sh on its own box.  OUT = output directory (under gpurun_out/); afterwards copy OUT/{insitu_*,g_*,pmc_traffic_*,j_*,d_*,h_*,q_*}
# into profiles/r5/ (tracked).  Order matters: the in-situ profiles first, so that the bench lines taken afterwards read them.
#   1. rocprofv3 --kernel-trace --stats of `bench.py --trace-run` per workload -> g_<w>_kernel_stats.csv + insitu_<tag>.json
#   2. PMC traffic passes per workload -> pmc_traffic_<tag>.{txt,json}
#   3. bench lines (they read 1 + 2 from profiles/r5/ - this script installs them there ON THE BOX before running them)
#   4. per-op tables, other configurations
OUT=$1; mkdir -p $OUT profiles/r5
R=$PWD
export TMPDIR=/tmp
W_F32=""; W_C3="--dtype bf16"; W_C5="--dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10"
python3 tools/conv_table.py > $OUT/h_conv_table_f32_config2.txt 2>/dev/null
python3 tools/conv_table.py --other > $OUT/h_other_ops_f32_config2.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 > $OUT/h_conv_table_bf16_config3.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --other > $OUT/h_other_ops_bf16_config3.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --img 128 --ksize 11 > $OUT/h_conv_table_bf16_config5.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --img 128 --ksize 11 --other > $OUT/h_other_ops_bf16_config5.txt 2>/dev/null
echo tables done
( for flags in "--loss wass --opt rmsprop" "--loss wass --opt rmsprop --dtype bf16" "--img 128 --ksize 11 --seq_len 16 --steps 10" "--plain" "--plain --dtype bf16" "--batch 64" "--dtype bf16 --batch 256 --steps 10"; do
    echo "== bench.py $flags"; python3 bench.py --no-cpu-baseline $flags 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step  conv', d['roofline']['achieved'], 'TFLOP/s (', d['roofline']['timing'][:12], ') dna', (d['roofline_dna'] or {}).get('frac'), '|', d['config']['workload'])"
  done ) > $OUT/q_other_configs.txt 2>&1
echo other configs done
rm -rf $OUT/prof_*/ $OUT/pmc_*/ $OUT/l2_*/
ls $OUT | head -60
