#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats` writes *_results.db on this ROCm).  Usage: rocpd_kernel_stats.py file.db [--skip N]"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in db.execute('pragma table_info(%s)' % disp)]
    scols = [r[1] for r in db.execute('pragma table_info(%s)' % sym)]
    name_col = 'kernel_name' if 'kernel_name' in scols else ('display_name' if 'display_name' in scols else scols[-1])
    q = ('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, disp, sym))
    rows = list(db.execute(q))
    stats = {}
    for name, st, en in rows:
        name = name.split('(')[0]
        stats.setdefault(name, []).append(en - st)
    total = sum(sum(v) for v in stats.values())
    print('%-44s %7s %12s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'share'))
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print('%-44s %7d %12.1f %12.2f %12.2f %12.2f %6.1f%%' % (name[:44], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3,
                                                                min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / total))
    if rows:
        span = rows[-1][2] - rows[0][1]
        print('# %d dispatches, kernel time %.1f us, first-start..last-end span %.1f us' % (len(rows), total / 1e3, span / 1e3))


if __name__ == '__main__':
    main()
