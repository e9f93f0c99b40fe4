#!/usr/bin/env python
"""Timeline of the LAST `n` kernel dispatches of a rocprofv3 --kernel-trace database (rocpd SQLite): start and end relative to the first of them, in us --
shows which kernels of different streams overlap.  Usage: kernel_timeline.py file.db [n=16]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scols = [r[1] for r in db.execute('pragma table_info(%s)' % sym)]
name_col = 'kernel_name' if 'kernel_name' in scols else 'display_name'
rows = list(db.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, disp, sym)))[-n:]
t0 = rows[0][1]
print('%-40s %12s %12s %10s' % ('kernel', 'start us', 'end us', 'us'))
for name, st, en in rows:
    print('%-40s %12.1f %12.1f %10.1f' % (name.split('(')[0][:40], (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3))
