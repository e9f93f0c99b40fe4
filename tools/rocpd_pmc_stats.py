#!/usr/bin/env python
"""Per-kernel average of a rocprofv3 --pmc counter from the rocpd SQLite database.  Usage: rocpd_pmc_stats.py file.db"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    info = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in db.execute('pragma table_info(%s)' % sym)]
    name_col = 'kernel_name' if 'kernel_name' in scols else 'display_name'
    q = ('select s.%s, i.name, e.value from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id '
         'join %s i on e.pmc_id = i.id' % (name_col, ev, disp, sym, info))
    acc = {}
    for kname, cname, val in db.execute(q):
        acc.setdefault((kname.split('(')[0], cname), []).append(val)
    print('%-44s %-12s %7s %14s %14s' % ('kernel', 'counter', 'calls', 'avg', 'max'))
    for (kname, cname), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print('%-44s %-12s %7d %14.1f %14.1f' % (kname[:44], cname, len(v), sum(v) / len(v), max(v)))


if __name__ == '__main__':
    main()
