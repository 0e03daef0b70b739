# rocprofv3 PMC passes over the bf16 conv probe (usage: bash tools/pmc_probe16.sh OUTDIR [probe args])
OUT=${1:-gpurun_out/pmc16}; shift
mkdir -p $OUT && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 tools/conv16_probe.py "$@" > $OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/b -- python3 tools/conv16_probe.py "$@" > $OUT/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/c -- python3 tools/conv16_probe.py "$@" > $OUT/c.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum TCC_TAG_STALL_sum --output-format csv -d $OUT/d -- python3 tools/conv16_probe.py "$@" > $OUT/d.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/e -- python3 tools/conv16_probe.py "$@" > $OUT/e.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0][-60:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        n = len(cnt[(k, c)]) or 1
        print('   %-28s %16.0f per launch' % (c, v / n))
PY
