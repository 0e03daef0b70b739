#!/usr/bin/env python
"""Per-kernel-family time of ONE real training step, in situ, from a rocprofv3 kernel trace of the bench itself.

bench.py's own per-op timing relaunches each op back to back (graph.profile_ops): every operand is cache-hot from the
previous identical launch, so it overstates what the op reaches inside the step.  This reduces

    rocprofv3 --kernel-trace --stats -d DIR -o NAME --output-format csv -- python3 bench.py --trace-run [flags] > LOG

to the time each kernel family takes per REAL step: `--trace-run` makes the bench execute nothing but training steps (no
rollout, no instrumented pass, no CPU baseline), and its JSON line says how many (`config.step_executions`: the eager /
capture / first-replay steps, the warm-up and every timed block), so family time per step = TotalDurationNs / steps.

    python tools/insitu_times.py DIR/..._kernel_stats.csv LOG > profiles/r5/insitu_<tag>.json

The JSON is what bench.py reads for `roofline` / `roofline_dna` / `op_ms_per_step` (like the PMC traffic summaries), and
profiles/r5/g_*_kernel_stats.csv is the very file it was reduced from: conv time per step = sum over the conv families of
TotalDurationNs / step_executions, by hand."""
import csv
import json
import re
import sys

# kernel-name prefix (after stripping `void `, namespaces and template arguments) -> family
FAMILIES = (
    ('conv', ('conv_mfma_f32', 'conv_pair_f32', 'conv_mfma_bf16', 'conv_pair_bf16', 'conv_glds_bf16', 'splitk_reduce', 'direct_fwd', 'direct_dgrad',
              'direct_wgrad', 'direct_pair', 'merge_dgrad_weights')),          # every contraction launch + its slab reductions
    ('bn', ('bn_',)),
    ('bias', ('bias_act', 'colsum_finalize')),
    ('dna_fwd', ('dna_fwd',)), ('dna_bwd', ('dna_bwd', 'dna_dbias')),
    ('optimizer', ('adam_k', 'rmsprop_k', 'clip_k', 'step_inc_k', 'weights_prepare', 'opt_prepare_k')),
    ('loss', ('frame_loss', 'l2norm_loss', 'sigmoid_ce', 'mean_loss', 'sqdiff', 'finalize_k', 'scalar_combine', 'psnr')),
    ('plumbing', ('copy_many', 'concat', 'slice', 'add_k', '__amd_rocclr')),
    ('rccl', ('ncclDevKernel', 'rccl')),
)


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::|acgconv::', '', name)
    # kernels with a bfloat16 template argument come through MANGLED (the profiler's demangler does not know DF16b):
    # _ZN12_GLOBAL__N_1<len><name>I<template args>E... / _ZN7acgconv<len><name>I...
    m = re.match(r'_ZN(?:12_GLOBAL__N_1|7acgconv)(?:12_GLOBAL__N_1)?(\d+)', name)
    if m:
        n0 = m.end()
        base = name[n0:n0 + int(m.group(1))]
        if base in ('dna_kernel', 'dna_rows_kernel'):       # ...ILi<K>E[Li<TY>E]Lb<0|1>E...: the first bool is BWD
            b = re.search(r'Lb([01])E', name[n0:])
            return 'dna_bwd' if (b and b.group(1) == '1') else 'dna_fwd'
        return base
    m = re.match(r'(dna_kernel|dna_rows_kernel)<([^>]*)>', name)      # forward / backward are one template
    if m:
        args = [a.strip() for a in m.group(2).split(',')]
        flag = args[2] if m.group(1) == 'dna_kernel' else args[1]     # <K, TY, BWD, ...> / <K, BWD, ...>
        return 'dna_bwd' if flag in ('true', '1', '(bool)1') else 'dna_fwd'
    return re.sub(r'[<(].*$', '', name)


def family(sname):
    for fam, prefixes in FAMILIES:
        if any(sname.startswith(p) for p in prefixes):
            return fam
    return 'other'


def bench_line(log_path):
    with open(log_path) as f:
        for line in f:
            line = line.strip()
            if line.startswith('{') and '"metric"' in line:
                return json.loads(line)
    raise SystemExit('no bench JSON line in %s' % log_path)


def main():
    stats_csv, log = sys.argv[1], sys.argv[2]
    line = bench_line(log)
    steps = line['config'].get('step_executions')
    if not steps or not line['config'].get('trace_run'):
        raise SystemExit('the bench line is not from a --trace-run (config.step_executions missing)')
    n_critic = line['config']['n_critic']
    fam, kernels = {}, {}
    with open(stats_csv) as f:
        for row in csv.DictReader(f):
            s = short(row['Name'])
            calls, tot = int(row['Calls']), float(row['TotalDurationNs'])
            k = kernels.setdefault(s, [0, 0.0])
            k[0] += calls
            k[1] += tot
            d = fam.setdefault(family(s), [0, 0.0])
            d[0] += calls
            d[1] += tot
    # consistency: the optimizer's step counter runs once per D step and once per G step
    inc = kernels.get('step_inc_k')
    if line['config'].get('opt', 'adam') == 'adam' and inc and inc[0] != steps * (n_critic + 1):
        raise SystemExit('step_inc_k ran %d times, expected %d steps x %d programs: the trace holds more than training steps'
                         % (inc[0], steps, n_critic + 1))
    out = {
        'source': 'rocprofv3 --kernel-trace --stats of `bench.py --trace-run` (tools/insitu_times.py); durations are the profiler\'s '
                  'per-dispatch begin -> end inside the replayed step graphs, operands as the step leaves them',
        'workload': line['config']['workload'], 'dtype': line['dtype'], 'step_executions': steps,
        # what the trace was taken with: bench.py refuses to quote this file for another library / ABI / call path
        'abi_version': line['config'].get('abi_version'), 'lib_sha16': line['config'].get('lib_sha16'), 'lookahead': line['config'].get('lookahead'),
        'bench_ms_per_step_under_profiler': line['ms_per_step'],
        'family_us_per_step': {k: round(v[1] / steps / 1e3, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
        'family_launches_per_step': {k: round(v[0] / steps, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
        'kernel_us_per_launch': {k: round(v[1] / v[0] / 1e3, 3) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1][1])},
        'kernel_launches_per_step': {k: round(v[0] / steps, 3) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1][1])},
    }
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write('\n')


if __name__ == '__main__':
    main()
