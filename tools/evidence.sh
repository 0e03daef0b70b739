#!/bin/bash
# One box: per-op tables, bench lines and rocprofv3 kernel stats of the three bench workloads.  OUT = output directory.
set -e
OUT=$1; mkdir -p $OUT
R=$PWD
python3 tools/conv_table.py > $OUT/h_conv_table_f32_config2.txt 2>/dev/null
python3 tools/conv_table.py --other > $OUT/h_other_ops_f32_config2.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 > $OUT/h_conv_table_bf16_config3.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --other > $OUT/h_other_ops_bf16_config3.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --img 128 --ksize 11 > $OUT/h_conv_table_bf16_config5.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --img 128 --ksize 11 --other > $OUT/h_other_ops_bf16_config5.txt 2>/dev/null
echo tables done
python3 bench.py > $OUT/d_bench_f32_config2.json 2>$OUT/bench_err.txt
python3 bench.py --dtype bf16 > $OUT/d_bench_bf16_config3_b32.json 2>>$OUT/bench_err.txt
python3 bench.py --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 > $OUT/d_bench_bf16_config5_geometry.json 2>>$OUT/bench_err.txt
echo bench done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_f32 -o f32 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/$OUT/prof_f32.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_bf16 -o bf16 --output-format csv -- python3 $R/bench.py --dtype bf16 --no-cpu-baseline > $R/$OUT/prof_bf16.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_c5 -o c5 --output-format csv -- python3 $R/bench.py --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --no-cpu-baseline > $R/$OUT/prof_c5.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -40
