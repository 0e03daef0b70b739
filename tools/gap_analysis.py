#!/usr/bin/env python
"""Share of a training step spent BETWEEN kernels: from a rocprofv3 --kernel-trace csv of bench.py, take the last `--steps` steps
(a step = the kernels between two consecutive pairs of adam_k launches), sum the kernel durations and the idle gaps between
consecutive kernels on the device timeline.  python tools/gap_analysis.py <kernel_trace.csv> [--steps 10]"""
import argparse, csv, sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=10)
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if 'adam_k' in r[2] or 'rmsprop_k' in r[2]]
    # two optimizer launches per step (D then G): step k ends at adam[2k+1]
    ends = adam[1::2]
    # the timed region of the bench = the window of `steps` consecutive steps with the smallest wall time (warm-up, the per-op
    # profiling pass and the evaluation rollout around it are slower)
    best = None
    for k in range(0, len(ends) - args.steps):
        w = rows[ends[k + args.steps]][1] - rows[ends[k]][1]
        if best is None or w < best[0]:
            best = (w, k)
    ends = ends[best[1]:best[1] + args.steps + 1]
    lo, hi = ends[0] + 1, ends[-1] + 1
    seg = rows[lo:hi]
    n = len(ends) - 1
    busy = sum(e - s for s, e, _ in seg)
    wall = seg[-1][1] - seg[0][0]
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
    overlap = sum(max(0, seg[i][1] - seg[i + 1][0]) for i in range(len(seg) - 1))
    gaps_sorted = sorted(gaps)
    print('%d steps, %d kernels per step' % (n, len(seg) // n))
    print('wall per step      %8.1f us' % (wall / n / 1e3))
    print('kernel time        %8.1f us per step (%.1f %% of wall; %.1f us of it overlapped with the next kernel)' % (busy / n / 1e3, 100.0 * busy / wall, overlap / n / 1e3))
    print('idle between kernels %6.1f us per step (%.1f %%): median gap %.2f us, p90 %.2f us, max %.1f us' % (
        sum(gaps) / n / 1e3, 100.0 * sum(gaps) / wall, gaps_sorted[len(gaps) // 2] / 1e3, gaps_sorted[int(len(gaps) * 0.9)] / 1e3, gaps_sorted[-1] / 1e3))
    big = sorted(((g, seg[i][2][:60], seg[i + 1][2][:60]) for i, g in enumerate(gaps)), reverse=True)[:5]
    for g, a, b in big:
        print('   gap %6.1f us between %s -> %s' % (g / 1e3, a, b))


if __name__ == '__main__':
    main()
