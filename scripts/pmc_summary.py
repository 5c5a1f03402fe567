#!/usr/bin/env python3
"""Folds rocprofv3 --pmc results (the *_results.db files under the given directories) into
{kernel: {counter: {mean_per_launch, launches}}} and prints it as JSON."""
import glob, json, os, sqlite3, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for root in sys.argv[1:]:
    for path in glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(path)
        cols = [r[1] for r in con.execute("pragma table_info(pmc_events)")]
        cname = "counter_name" if "counter_name" in cols else "pmc_name"
        vname = "counter_value" if "counter_value" in cols else "value"
        q = f"select name, dispatch_id, {cname}, sum({vname}) from pmc_events group by 1,2,3"
        for kernel, _, counter, value in con.execute(q):
            a = acc[kernel][counter]
            a[0] += value
            a[1] += 1
out = {k: {c: {"mean_per_launch": a[0] / a[1], "launches": a[1]} for c, a in cs.items()}
       for k, cs in acc.items()}
json.dump(out, sys.stdout, indent=1)
