#!/bin/bash
# Round-5 evidence on ONE box.  OUT = output directory (under gpurun_out/); afterwards copy OUT/{insitu_*,g_*,pmc_traffic_*,j_*,d_*,h_*,q_*}
# into profiles/r5/ (tracked).  Order matters: the in-situ profiles first, so that the bench lines taken afterwards read them.
#   1. rocprofv3 --kernel-trace --stats of `bench.py --trace-run` per workload -> g_<w>_kernel_stats.csv + insitu_<tag>.json
#   2. PMC traffic passes per workload -> pmc_traffic_<tag>.{txt,json}
#   3. bench lines (they read 1 + 2 from profiles/r5/ - this script installs them there ON THE BOX before running them)
#   4. per-op tables, other configurations
OUT=$1; mkdir -p $OUT profiles/r5
R=$PWD
export TMPDIR=/tmp
W_F32=""; W_C3="--dtype bf16"; W_C5="--dtype bf16 --img 128 --ksize 11 --seq_len 16 --steps 10"
trace() {   # name tag flags...
  local name=$1 tag=$2; shift 2
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$name -o $name --output-format csv -- python3 $R/bench.py --trace-run "$@" > $R/$OUT/prof_$name.log 2>$R/$OUT/prof_$name.err )
  local stats=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  cp $stats $OUT/g_${name}_kernel_stats.csv
  python3 tools/insitu_times.py $stats $OUT/prof_$name.log > $OUT/insitu_$tag.json && cp $OUT/insitu_$tag.json profiles/r5/
  find $OUT/prof_$name -name "*kernel_trace.csv" -delete
}
trace f32_config2 f32_b32_s64_k5 $W_F32
trace bf16_config3 bf16_b32_s64_k5 $W_C3
trace bf16_config5 bf16_b32_s128_k11 $W_C5
echo insitu done
pmc() {     # tag flags...
  local tag=$1; shift
  bash tools/pmc_traffic.sh $OUT/pmc_$tag "$@" > /dev/null 2>&1
  cp $OUT/pmc_$tag/summary.txt $OUT/pmc_traffic_$tag.txt; cp $OUT/pmc_$tag/summary.json $OUT/pmc_traffic_$tag.json
  cp $OUT/pmc_traffic_$tag.json $OUT/pmc_traffic_$tag.txt profiles/r5/
}
pmc f32_b32_s64_k5 $W_F32
pmc bf16_b32_s64_k5 $W_C3
pmc bf16_b32_s128_k11 --dtype bf16 --img 128 --ksize 11 --seq_len 16
l2() {      # tag flags... : L2 (TCC) hits / misses per kernel family of the step - where the conv kernels' memory-side traffic comes from
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/l2_$tag -- python3 bench.py --steps 4 --warmup 1 --min-seconds 0 --no-cpu-baseline --profile-repeats 1 "$@" > $OUT/l2_$tag.log 2>&1
  python3 - $OUT/l2_$tag > $OUT/j_l2_hit_miss_$tag.txt <<'PY'
import csv, glob, os, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(path)):
        n = re.sub(r'^void |\(anonymous namespace\)::|acgconv::', '', row['Kernel_Name'])
        m = re.match(r'_ZN(?:12_GLOBAL__N_1|7acgconv)(?:12_GLOBAL__N_1)?(\d+)', n)
        k = n[m.end():m.end() + int(m.group(1))] if m else re.sub(r'[<(].*$', '', n)
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); disp[k].add(row['Dispatch_Id'])
print('# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum over bench.py (per launch; requests are 128-byte lines)')
print('# %-30s %9s %12s %12s %7s %10s' % ('kernel', 'launches', 'TCC_REQ', 'TCC_MISS', 'miss %', 'miss MB'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('TCC_MISS_sum', 0)):
    n = len(disp[k]); req, miss = v.get('TCC_REQ_sum', 0) / n, v.get('TCC_MISS_sum', 0) / n
    if req > 1000: print('%-32s %9d %12.0f %12.0f %7.1f %10.2f' % (k[:32], n, req, miss, 100 * miss / max(req, 1), miss * 128 / 1e6))
PY
}
l2 f32_b32_s64_k5 $W_F32
l2 bf16_b32_s128_k11 --dtype bf16 --img 128 --ksize 11 --seq_len 16
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
echo pmc done
python3 bench.py $W_F32 > $OUT/d_bench_f32_config2.json 2>$OUT/bench_err.txt
python3 bench.py $W_C3 > $OUT/d_bench_bf16_config3_b32.json 2>>$OUT/bench_err.txt
python3 bench.py $W_C5 > $OUT/d_bench_bf16_config5_geometry.json 2>>$OUT/bench_err.txt
echo bench done
rm -rf $OUT/prof_*/ $OUT/pmc_*/ $OUT/l2_*/
ls $OUT | head -60
