#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch.

    python scripts/pmc_summary.py <fetch_dir> <write_dir> <out.csv>

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KiB, collected in
separate passes, and on gfx950 FETCH_SIZE reports half of the bytes of coalesced reads, so
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (averaged over the dispatches of
a kernel in the profiled run).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            k = r['Kernel_Name']
            acc[k][0] += float(r['Counter_Value'])
            acc[k][1] += 1
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch = load(fdir, 'FETCH_SIZE')
    write = load(wdir, 'WRITE_SIZE')
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f = fetch[k][0] / fetch[k][1] if fetch[k][1] else 0.0
        w = write[k][0] / write[k][1] if write[k][1] else 0.0
        rows.append((k, max(fetch[k][1], write[k][1]), f, w, int((2 * f + w) * 1024)))
    rows.sort(key=lambda r: -r[4])
    with open(out, 'w', newline='') as fo:
        wr = csv.writer(fo)
        wr.writerow(['Kernel', 'dispatches', 'FETCH_SIZE_KiB_avg(raw)', 'WRITE_SIZE_KiB_avg',
                     'HBM_bytes_per_launch=(2*FETCH+WRITE)*1024'])
        for r in rows:
            wr.writerow([r[0], r[1], '%.1f' % r[2], '%.1f' % r[3], r[4]])


if __name__ == '__main__':
    main()
