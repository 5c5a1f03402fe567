"""Per-API call count, average and total host time from a rocprofv3 --hip-trace results database."""
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0])
rows = db.execute("select name, count(*), avg(end - start), sum(end - start) from regions "
                  "group by name order by 4 desc").fetchall()
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"  {r[0][:44]:46s} x{r[1]:6d} {r[2] / 1e3:9.1f} us each {r[3] / 1e6:9.3f} ms total")
