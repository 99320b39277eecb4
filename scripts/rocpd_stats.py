"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) like `--stats`: per-kernel calls, total, avg.
usage: python scripts/rocpd_stats.py <db> [rows] [kernel-name substring: also print that kernel's launches grouped by workgroup count --
the per-kernel average mixes every launch size of the run (full launch sets, 2-chunk sets, reference trajectories); bench.py's
roofline.launch_kinds lists the same groups from its live HIP-event measurement]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}")
for n, c, t, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)[:90]
    print(f"{n:90s} {c:7d} {t/1e6:10.2f} {t/c/1e3:9.2f} {mn/1e3:8.2f} {mx/1e3:9.2f} {100*t/tot:6.2f}")
print(f"TOTAL kernel time {tot/1e6:.2f} ms")
# GPU occupancy in time: union of all kernel intervals over the span from the first to the last kernel
iv = cur.execute(f"select start, end from {kd} order by start").fetchall()
if iv:
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > ce:
            busy += ce - cs; cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    span = max(b for _, b in iv) - iv[0][0]
    print(f"kernel-interval union {busy/1e6:.2f} ms of {span/1e6:.2f} ms span ({100.0*busy/span:.1f} % busy); sum of durations / union = {tot/busy:.3f} (stream overlap)")

if len(sys.argv) > 3:
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    gx = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "grid_size")]
    wx = [c for c in cols if c.lower() in ("workgroup_size_x", "workgroup_x", "workgroup_size")]
    if gx:
        wg = f"d.{gx[0]} / d.{wx[0]}" if wx else f"d.{gx[0]}"
        q = (f"select s.kernel_name, {wg} as wgs, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id "
             f"where s.kernel_name like ? group by s.kernel_name, wgs order by 1, 2 desc")
        print(f"launches of kernels matching '{sys.argv[3]}' by workgroup count:")
        for n, w, c, t in cur.execute(q, (f"%{sys.argv[3]}%",)).fetchall():
            n = re.sub(r"\(anonymous namespace\)::", "", n)[:70]
            print(f"  {n:70s} workgroups {int(w):7d} calls {c:6d} avg_us {t/c/1e3:9.2f}")
    else:
        print("(no grid-size column in this rocpd schema:", cols, ")")
