import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, stream_id, queue_id, start, end from kernels order by start"))
by = collections.OrderedDict()
for n, s, q, a, b in rows:
    by.setdefault((s, q), []).append(n[:50])
for (s, q), v in by.items():
    print(f"stream {s} queue {q}: {len(v)} kernels, e.g. {v[0]}")
