#!/bin/bash
# Round-3 evidence on ONE box: per-op tables, bench lines, rocprofv3 kernel stats and PMC traffic of the three bench workloads,
# plus the other BASELINE configurations.  OUT = output directory (under gpurun_out/).
OUT=$1; mkdir -p $OUT
R=$PWD
python3 bench.py > $OUT/d_bench_f32_config2.json 2>$OUT/bench_err.txt
python3 bench.py --dtype bf16 > $OUT/d_bench_bf16_config3_b32.json 2>>$OUT/bench_err.txt
python3 bench.py --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 > $OUT/d_bench_bf16_config5_geometry.json 2>>$OUT/bench_err.txt
echo bench done
python3 tools/conv_table.py > $OUT/h_conv_table_f32_config2.txt 2>/dev/null
python3 tools/conv_table.py --other > $OUT/h_other_ops_f32_config2.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 > $OUT/h_conv_table_bf16_config3.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --other > $OUT/h_other_ops_bf16_config3.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --img 128 --ksize 11 > $OUT/h_conv_table_bf16_config5.txt 2>/dev/null
python3 tools/conv_table.py --dtype bf16 --img 128 --ksize 11 --other > $OUT/h_other_ops_bf16_config5.txt 2>/dev/null
echo tables done
( for flags in "--loss wass --opt rmsprop" "--loss wass --opt rmsprop --dtype bf16" "--img 128 --ksize 11 --seq_len 16 --steps 10" "--plain" "--plain --dtype bf16" "--batch 64" "--dtype bf16 --batch 256 --steps 10"; do
    echo "== bench.py $flags"; python3 bench.py --no-cpu-baseline $flags 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step  conv', d['roofline']['achieved'], 'TFLOP/s  dna', (d['roofline_dna'] or {}).get('frac'), '|', d['config']['workload'])"
  done ) > $OUT/q_other_configs.txt 2>&1
echo other configs done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_f32 -o f32 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/$OUT/prof_f32.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_bf16 -o bf16 --output-format csv -- python3 $R/bench.py --dtype bf16 --no-cpu-baseline > $R/$OUT/prof_bf16.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_c5 -o c5 --output-format csv -- python3 $R/bench.py --dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10 --no-cpu-baseline > $R/$OUT/prof_c5.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -delete
echo rocprof done
bash tools/pmc_traffic.sh $OUT/pmc_f32 > /dev/null 2>&1
bash tools/pmc_traffic.sh $OUT/pmc_bf16_c3 --dtype bf16 > /dev/null 2>&1
bash tools/pmc_traffic.sh $OUT/pmc_bf16_c5 --dtype bf16 --img 128 --ksize 11 --seq_len 16 > /dev/null 2>&1
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
echo pmc done
ls -R $OUT | head -60
