# rocprofv3 PMC passes over the BatchNorm kernels at BASELINE sizes (usage: bash tools/pmc_bn.sh OUTDIR [bench_bn.py flags])
# Four passes: SQ (issue / wait / occupancy), FETCH_SIZE, WRITE_SIZE, TCC hit / miss - the TCC slots do not fit together
# (MI355X_MICROARCH.md, rocprofv3 PMC slots).  The program itself follows `--` (no env / bash -c hop).
OUT=${1:-gpurun_out/pmc_bn}; shift
mkdir -p $OUT && export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/$OUT/a -- python3 $R/tools/bench_bn.py "$@" > $R/$OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/b -- python3 $R/tools/bench_bn.py "$@" > $R/$OUT/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/c -- python3 $R/tools/bench_bn.py "$@" > $R/$OUT/c.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/d -- python3 $R/tools/bench_bn.py "$@" > $R/$OUT/d.log 2>&1
cd $R
python3 - "$OUT" "$*" <<'PY'
import csv, glob, collections, re, sys
out, flags = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(list)
def short(k):
    k = re.sub(r'^void ', '', k); k = re.sub(r'\(anonymous namespace\)::', '', k)
    m = re.match(r'(bn_[a-z_]+)<([^>]*)>', k)
    if not m: return None
    a = [x.strip() for x in m.group(2).split(',')]
    keep = [x for x in a if x in ('float', '__bf16', '__hip_bfloat16') or x.isdigit()]
    return m.group(1) + '<' + ','.join(x.replace('__hip_bfloat16', 'bf16').replace('__bf16', 'bf16').replace('float', 'f32') for x in keep) + '>'
for path in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k = short(row['Kernel_Name'])
        if not k: continue
        k = '%s grid %s wg %s' % (k, row.get('Grid_Size', '?'), row.get('Workgroup_Size', '?'))
        acc[k][row['Counter_Name']] += float(row['Counter_Value']); n[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
for path in glob.glob(out + '/a/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k = short(row['Kernel_Name'])
        if k: dur['%s grid %s wg %s' % (k, row.get('Grid_Size', '?'), row.get('Workgroup_Size', '?'))].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
with open(out + '/summary.txt', 'w') as f:
    f.write('# rocprofv3 --pmc over tools/bench_bn.py %s; per-dispatch means per (kernel, grid); SQ cycle counters are quad-cycles summed over waves;\n' % flags)
    f.write('# FETCH_SIZE / WRITE_SIZE in KB (FETCH x2 = bytes of a wide streaming read on gfx950); durations are under the profiler\n')
    for k in sorted(acc):
        a = acc[k]
        def per(c): return a[c] / max(1, len(n[(k, c)])) if c in a else float('nan')
        f.write(k + '\n')
        d = sorted(dur.get(k, [0]))
        us = d[len(d) // 2] / 1e3
        f.write('  dispatches %d, median %.2f us\n' % (len(d), us))
        if 'SQ_WAVE_CYCLES' in a:
            wc = per('SQ_WAVE_CYCLES')
            f.write('  waves %.0f; of wave cycles: waiting (s_waitcnt / barrier) %.1f %%, issue stall %.1f %%, issuing %.1f %%; VALU / wave %.0f, VMEM reads / wave %.1f\n' % (
                per('SQ_WAVES'), 100 * per('SQ_WAIT_ANY') / wc, 100 * per('SQ_WAIT_INST_ANY') / wc, 100 * per('SQ_ACTIVE_INST_ANY') / wc,
                per('SQ_INSTS_VALU') / max(per('SQ_WAVES'), 1), per('SQ_INSTS_VMEM_RD') / max(per('SQ_WAVES'), 1)))
            f.write('  mean resident waves per CU while busy: %.1f (wave quad-cycles / busy cycles x 4 / 256 CUs)\n' % (wc * 4 / max(per('SQ_BUSY_CYCLES'), 1) / 256 * 8))
        if 'FETCH_SIZE' in a:
            fb, wb = 2 * per('FETCH_SIZE') * 1024, per('WRITE_SIZE') * 1024
            f.write('  HBM-side: fetch %.2f MB (x2 corrected), write %.2f MB' % (fb / 1e6, wb / 1e6))
            if us > 0: f.write(' -> %.0f GB/s over the profiled duration' % ((fb + wb) / us / 1e3))
            f.write('\n')
        if 'TCC_HIT_sum' in a:
            h, m = per('TCC_HIT_sum'), per('TCC_MISS_sum')
            f.write('  L2: hit %.0f miss %.0f -> hit rate %.1f %%\n' % (h, m, 100 * h / max(h + m, 1)))
print(open(out + '/summary.txt').read())
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
