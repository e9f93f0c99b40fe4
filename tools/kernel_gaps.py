#!/usr/bin/env python
"""Gaps between consecutive kernel dispatches in a rocprofv3 --kernel-trace database (rocpd SQLite): for every ordered pair of kernel names that
follow each other, the number of occurrences and the mean / minimum idle time between the end of the first and the start of the second.
Usage: kernel_gaps.py file.db"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in db.execute('pragma table_info(%s)' % sym)]
    name_col = 'kernel_name' if 'kernel_name' in scols else 'display_name'
    rows = list(db.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, disp, sym)))
    gaps = defaultdict(list)
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        gaps[(n0.split('(')[0][:34], n1.split('(')[0][:34])].append((s1 - e0) / 1e3)
    print('%-36s %-36s %6s %10s %10s' % ('kernel', 'followed by', 'count', 'mean us', 'min us'))
    for (a, b), g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:14]:
        print('%-36s %-36s %6d %10.2f %10.2f' % (a, b, len(g), sum(g) / len(g), min(g)))


if __name__ == '__main__':
    main()
