"""Durations (us) of every dispatch of the kernels whose name contains the given substring, in
launch order, from a rocprofv3 results database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute(
    f"select (d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id "
    f"where s.kernel_name like ? order by d.start", ("%" + sys.argv[2] + "%",)).fetchall()
print(sys.argv[2], len(rows), " ".join(f"{r[0]:.1f}" for r in rows))
