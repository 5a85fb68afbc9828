#!/usr/bin/env python
"""Print per-dispatch PMC counters from a rocprofv3 rocpd sqlite db. Usage: pmc_summary.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print(cols)
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = (r[ix.get('dispatch_id', 0)], r[ix['kernel_name']] if 'kernel_name' in ix else '')
    agg[k][r[ix['counter_name']]].append(r[ix['value']])
for (did, kn), d in sorted(agg.items()):
    print(did, kn[:60], {c: f"{sum(v):.4g}" for c, v in d.items()})
