"""Aggregate the margin records of repeated GPU suite runs (tests/_margins.py, GC_TEST_MARGINS): per (test, check) the worst
value / bar over all runs; prints every check that used more than half of its bar.  usage: python scripts/margins_summary.py f1.jsonl f2.jsonl ..."""
import collections
import json
import sys

worst = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0, 1e300])
for path in sys.argv[1:]:
    for line in open(path):
        r = json.loads(line)
        ratio = r["value"] / r["bar"] if r["bar"] else (0.0 if r["value"] == 0 else float("inf"))
        w = worst[(r["test"], r["check"][:90])]
        w[3] += 1
        w[4] = min(w[4], ratio)
        if ratio >= w[0]:
            w[0], w[1], w[2] = ratio, r["value"], r["bar"]
print(f"{len(worst)} distinct numeric checks over {len(sys.argv) - 1} suite runs; checks whose worst run used > 50 % of the bar:")
print(f"{'worst':>7s} {'best':>7s} {'value':>10s} {'bar':>10s} {'n':>3s}  check  [test]")
for (t, c), (ratio, v, b, n, lo) in sorted(worst.items(), key=lambda kv: -kv[1][0]):
    if ratio > 0.5:
        print(f"x{ratio:6.3f} x{lo:6.3f} {v:10.4g} {b:10.4g} {n:3d}  {c}  [{t}]")
