#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) results.db into the per-kernel --stats summary as text/CSV.
usage: rocpd_summary.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(top_kernels)")]
rows = db.execute("select * from top_kernels").fetchall()
lines = [','.join(cols)] + [','.join(str(x) for x in r) for r in rows]
txt = '\n'.join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + '\n')
print(txt)
