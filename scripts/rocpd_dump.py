"""Dump every kernel dispatch of a rocprofv3 rocpd SQLite result as `start_ns end_ns queue stream kernel-name-prefix` lines (offline gap / overlap analysis).
usage: python scripts/rocpd_dump.py <db> <out.txt>"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
q = "d.queue_id" if "queue_id" in cols else "0"
st = "d.stream_id" if "stream_id" in cols else "0"
rows = cur.execute(f"select d.start, d.end, {q}, {st}, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
t0 = rows[0][0] if rows else 0
with open(sys.argv[2], "w") as f:
    for a, b, qq, ss, n in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)[:28]
        f.write(f"{a - t0} {b - t0} {qq} {ss} {n}\n")
print(len(rows), "dispatches", cols)
