"""HBM traffic per dispatch of the kernels matching a regex: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes
(MI355X_MICROARCH.md), corrected with the factors scripts/pmc_traffic.py measures on this ROCm (FETCH_SIZE x 2, WRITE_SIZE x 1; KiB).
usage (GPU box): python scripts/pmc_kernel_traffic.py '<kernel regex>' <out.json> -- <command ...>"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

rx = re.compile(sys.argv[1]); outp = sys.argv[2]; cmd = sys.argv[sys.argv.index("--") + 1:]
os.environ["TMPDIR"] = "/tmp"
CORR = {"FETCH_SIZE": 2.0, "WRITE_SIZE": 1.0}
res = collections.defaultdict(dict)
for counter in CORR:
    d = f"/tmp/pmck_{os.getpid()}_{counter}"
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                       cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(f"# {counter}: no counter output (rc={r.returncode})\n" + r.stdout[-800:]); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter and rx.search(row["Kernel_Name"]):
            k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])[:80]
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        res[k][counter + "_MB_per_dispatch"] = round(v * 1024.0 * CORR[counter] / n / 1e6, 2)
        res[k]["dispatches"] = n
for k, v in res.items():
    if "FETCH_SIZE_MB_per_dispatch" in v and "WRITE_SIZE_MB_per_dispatch" in v:
        v["traffic_MB_per_dispatch"] = round(v["FETCH_SIZE_MB_per_dispatch"] + v["WRITE_SIZE_MB_per_dispatch"], 2)
json.dump({"kernels": res, "command": " ".join(cmd), "correction": CORR}, open(outp, "w"), indent=1)
print(json.dumps(res, indent=1))
