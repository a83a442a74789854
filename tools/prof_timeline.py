#!/usr/bin/env python3
"""Kernel timeline / per-kernel statistics from a rocprofv3 results .db (rocprofv3 --kernel-trace -d DIR -o NAME)."""
import collections, glob, sqlite3, statistics, sys
path = sys.argv[1]
n_tail = int(sys.argv[2]) if len(sys.argv) > 2 else 0
db = sqlite3.connect(glob.glob(path)[0])
cur = db.cursor()
suf = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
ks = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from rocpd_info_kernel_symbol{suf}")}
rows = list(cur.execute(f"select kernel_id, start, end, queue_id from rocpd_kernel_dispatch{suf} order by start"))
agg = collections.defaultdict(list)
for k, s, e, q in rows:
    agg[ks[k][:48]].append((e - s) / 1e3)
print(f"{'kernel':50s} {'calls':>6s} {'mean us':>9s} {'median':>9s} {'max':>9s} {'total ms':>9s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:50s} {len(v):6d} {statistics.mean(v):9.1f} {statistics.median(v):9.1f} {max(v):9.1f} {sum(v) / 1e3:9.2f}")
if n_tail:
    t0 = rows[-n_tail][1]
    for k, s, e, q in rows[-n_tail:]:
        print("%9.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, ks[k][:32]))
