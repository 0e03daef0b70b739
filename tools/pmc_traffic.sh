# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as the TCC slots require) of every kernel of the bench step.
# usage: bash tools/pmc_traffic.sh OUTDIR [bench.py flags]
OUT=${1:-gpurun_out/pmc_traffic}; shift
mkdir -p $OUT && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 4 --warmup 1 --min-seconds 0 --no-cpu-baseline --profile-repeats 1 "$@" > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 4 --warmup 1 --min-seconds 0 --no-cpu-baseline --profile-repeats 1 "$@" > $OUT/write.log 2>&1
python3 tools/pmc_summarise.py $OUT > $OUT/summary.txt 2> $OUT/summary.json
cat $OUT/summary.txt | head -30
