# rocprofv3 PMC passes over the DNA stencil kernels (usage: bash tools/pmc_dna.sh OUTDIR [ksize] [dtype] [img])
# Two passes (8 SQ counters each); the program itself follows `--` (no env / bash -c hop: the box refuses an exec after GPU init).
OUT=${1:-gpurun_out/pmc_dna}; K=${2:-11}; DT=${3:-bf16}; IMG=${4:-128}
mkdir -p $OUT && export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/$OUT/a -- python3 $R/tools/bench_dna.py --img $IMG --ksize $K --dtype $DT --batches 32 --reps 5 > $R/$OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/$OUT/b -- python3 $R/tools/bench_dna.py --img $IMG --ksize $K --dtype $DT --batches 32 --reps 5 > $R/$OUT/b.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for path in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name']
        if 'dna' not in k: continue
        k = ('dna_bwd' if 'Lb1E' in k else 'dna_fwd') if 'dna_rows' in k or 'dna_kernel' in k else k[:30]
        acc[k][row['Counter_Name']] += float(row['Counter_Value']); n[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
with open('$OUT/summary.txt', 'w') as f:
    f.write('# rocprofv3 --pmc, tools/bench_dna.py --img $IMG --ksize $K --dtype $DT --batches 32; per-dispatch means; SQ cycle counters are quad-cycles summed over waves\n')
    for k in sorted(acc):
        f.write(k + '\n')
        for c in sorted(acc[k]):
            f.write('  %-24s %16.0f\n' % (c, acc[k][c] / max(1, len(n[(k, c)]))))
        a = acc[k]
        def per(c): return a[c] / max(1, len(n[(k, c)]))
        if 'SQ_WAVE_CYCLES' in a:
            wc = per('SQ_WAVE_CYCLES')
            f.write('  -> of wave cycles: waiting (s_waitcnt / barrier) %.1f %%, issue stall %.1f %%, issuing %.1f %%; VALU-issuing %.1f %%\n' % (
                100 * per('SQ_WAIT_ANY') / wc, 100 * per('SQ_WAIT_INST_ANY') / wc, 100 * per('SQ_ACTIVE_INST_ANY') / wc, 100 * per('SQ_ACTIVE_INST_VALU') / wc))
            f.write('  -> VALU instructions per wave %.0f; busy cycles %.0f\n' % (per('SQ_INSTS_VALU') / max(per('SQ_WAVES'), 1), per('SQ_BUSY_CYCLES')))
print(open('$OUT/summary.txt').read())
PY
