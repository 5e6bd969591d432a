#!/usr/bin/env python3
"""Timeline of the last `n` GPU activities (kernels + memory copies) of a rocprofv3 rocpd database:
    python tools/timeline.py trace.db [n]
start / end in microseconds relative to the first listed activity."""
import sqlite3, sys
db = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
con = sqlite3.connect(db); cur = con.cursor()
rows = []
try:
    for name, s, e in cur.execute("select name, start, end from kernels"): rows.append((s, e, "K " + name.split("(")[0].replace("void ", "")[-60:]))
except sqlite3.Error as ex: print("kernels:", ex)
try:
    for name, s, e, size in cur.execute("select name, start, end, size from memory_copies"): rows.append((s, e, f"C {name} {size} B"))
except sqlite3.Error as ex: print("memory_copies:", ex)
rows.sort(); rows = rows[-n:]
t0 = rows[0][0] if rows else 0
for s, e, what in rows: print(f"{(s-t0)/1e3:10.1f} {(e-t0)/1e3:10.1f} {(e-s)/1e3:8.1f}  {what}")
