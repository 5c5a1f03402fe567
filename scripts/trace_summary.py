"""Per-kernel call count, average and total duration from a rocprofv3 results database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute(
    f"select s.kernel_name,count(*),avg(d.end-d.start),sum(d.end-d.start) from {kd} d "
    f"join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc").fetchall()
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(r[0][:90].ljust(90), r[1], round(r[2] / 1e3, 1), round(r[3] / 1e3, 1))
