#!/usr/bin/env python
"""Per-kernel cycle accounting from the SQ pass of tools/pmc_mfma.sh (rocprofv3 --pmc over bench.py --trace-run).

Units as /opt/skills/guides/MI355X_MICROARCH.md states them: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per v_mfma_f32_32x32x2_f32
and SIMD, 32 per v_mfma_f32_32x32x16_bf16); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves
(WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).  Two MFMA figures per kernel:
  "of wave time"  = MFMA_BUSY / (4 x WAVE_CYCLES): matrix-pipe cycles per cycle of resident-wave lifetime (with w waves sharing a SIMD
                    the pipe's duty cycle while they are resident is w times this);
  "of the kernel" = MFMA_BUSY / (kernel duration x 2.4 GHz x 4 SIMDs x 256 CUs), duration from the kernel trace of the same pass: the
                    share of ALL the machine's matrix-pipe cycles between the kernel's start and end - tile quantisation (CUs without a
                    block), launch ramp and tail included.  It is the in-kernel counterpart of the roofline fraction."""
import csv
import glob
import os
import sys

from pmc_summarise import load, short

CUS, SIMDS, GHZ = 256, 4, 2.4


def durations(root):
    per = {}
    for path in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                d = per.setdefault(short(row['Kernel_Name']), [0, 0.0])
                d[0] += 1
                d[1] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    return per


def main():
    root = sys.argv[1]
    names = ('SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_MFMA')
    per = {n: load(root, n) for n in names}
    dur = durations(root)
    total = sum(v[1] for v in dur.values()) or 1.0
    kernels = sorted(dur, key=lambda k: -dur[k][1])
    print('%-30s %8s %10s %10s %15s %15s   %s' % ('kernel', 'launches', '% of time', 'us/launch', 'MFMA of wave t.', 'MFMA of kernel', 'wave time: issuing / issue-stalled / parked %'))
    for k in kernels:
        n, ns = dur[k]
        if ns / total < 0.004:
            continue
        g = lambda c: per[c].get(k, {'sum': 0.0})['sum']      # noqa: E731
        wc = g('SQ_WAVE_CYCLES') or 1.0
        mfma = g('SQ_VALU_MFMA_BUSY_CYCLES')
        print('%-30s %8d %9.1f%% %10.1f %14.1f%% %14.1f%%   %5.1f / %5.1f / %5.1f' %
              (k, n, 100 * ns / total, ns / n / 1e3, 100 * mfma / (4 * wc), 100 * mfma / (ns * GHZ * SIMDS * CUS),
               100 * g('SQ_ACTIVE_INST_ANY') / wc, 100 * g('SQ_WAIT_INST_ANY') / wc, 100 * g('SQ_WAIT_ANY') / wc))
    conv = [k for k in kernels if k.startswith(('conv_mfma', 'conv_pair', 'conv_glds'))]
    if conv:
        ns = sum(dur[k][1] for k in conv)
        mfma = sum(per['SQ_VALU_MFMA_BUSY_CYCLES'].get(k, {'sum': 0.0})['sum'] for k in conv)
        wc = sum(per['SQ_WAVE_CYCLES'].get(k, {'sum': 0.0})['sum'] for k in conv) or 1.0
        print('# conv kernels together: %.1f %% of the kernel time of this pass; matrix pipe busy %.1f %% of their waves\' lifetime, %.1f %% of the machine\'s '
              'pipe cycles over their duration' % (100 * ns / total, 100 * mfma / (4 * wc), 100 * mfma / (ns * GHZ * SIMDS * CUS)))
        print('# (durations are those of the counter pass itself, where every dispatch is serialised and runs a little slower than in the plain trace)')


if __name__ == '__main__':
    main()
