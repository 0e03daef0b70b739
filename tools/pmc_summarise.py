#!/usr/bin/env python
"""Per-kernel HBM-side traffic table from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh.
  python tools/pmc_summarise.py gpurun_out/pmc_traffic > profiles/rN/..._pmc_traffic.txt   (also writes a .json beside stdout's data)
Counter unit: KB.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::|acgconv::', '', name)
    name = re.sub(r'\(.*$', '', name)                       # drop the argument list
    if name.startswith('conv_mfma_f32'):
        return 'conv_mfma_f32'
    return name[:34]


def load(dirname, counter):
    per = {}
    for path in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get('Counter_Name') != counter:
                    continue
                k = short(row['Kernel_Name'])
                d = per.setdefault(k, {'disp': set(), 'sum': 0.0})
                d['disp'].add(row['Dispatch_Id'])
                d['sum'] += float(row['Counter_Value'])
    return per


def main():
    root = sys.argv[1]
    fetch, write = load(os.path.join(root, 'fetch'), 'FETCH_SIZE'), load(os.path.join(root, 'write'), 'WRITE_SIZE')
    rows = []
    for k in sorted(set(fetch) | set(write)):
        nf = len(fetch.get(k, {'disp': ()})['disp']) or 1
        nw = len(write.get(k, {'disp': ()})['disp']) or 1
        f_mb = 2.0 * fetch.get(k, {'sum': 0.0})['sum'] * 1024 / 1e6 / nf      # KB -> MB, x2 gfx950 correction
        w_mb = write.get(k, {'sum': 0.0})['sum'] * 1024 / 1e6 / nw
        rows.append((k, max(nf, nw), f_mb, w_mb))
    rows.sort(key=lambda r: -(r[1] * (r[2] + r[3])))
    print('%-36s %8s   %-30s %-17s %s' % ('kernel', 'launches', 'fetch MB/launch (x2 corrected)', 'write MB/launch', 'total MB/launch'))
    for k, n, f_mb, w_mb in rows:
        print('%-36s %8d   %12.2f %26.2f %16.2f' % (k, n, f_mb, w_mb, f_mb + w_mb))
    out = {}
    for k, n, f_mb, w_mb in rows:
        if k in ('conv_mfma_f32', 'splitk_reduce', 'splitk_reduce_many'):
            out[k] = {'launches': n, 'fetch_bytes_per_launch': round(f_mb * 1e6), 'write_bytes_per_launch': round(w_mb * 1e6)}
    sys.stderr.write(json.dumps(out, indent=1) + '\n')


if __name__ == '__main__':
    main()
