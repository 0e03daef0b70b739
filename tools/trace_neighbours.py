"""Which kernels surround a given kernel in a rocprofv3 --kernel-trace CSV?  python tools/trace_neighbours.py trace.csv NAME_SUBSTRING"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]
ctx = collections.Counter()
dur = []
for i, r in enumerate(rows):
    if pat in r['Kernel_Name']:
        prev = rows[i - 1]['Kernel_Name'][:60] if i else '-'
        nxt = rows[i + 1]['Kernel_Name'][:60] if i + 1 < len(rows) else '-'
        ctx[(prev, nxt)] += 1
        dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print(len(dur), 'dispatches of', pat, 'mean ns', sum(dur) / max(len(dur), 1))
for (p, n), c in ctx.most_common(12):
    print('%5d  after %-60s before %s' % (c, p, n))
