"""(TCC / FETCH_SIZE groups are deliberately not collected: that pass did not finish within 15 minutes on this pool.)
Collect SQ counters for the kernels matching a regex (rocprofv3 --pmc, one pass per counter group, kernel trace only).
usage (on the GPU box): python scripts/pmc.py '<kernel regex>' -- <command ...>"""
import csv, glob, os, re, subprocess, sys, collections
SETS = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
    "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32",
    "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL",
]
rx = re.compile(sys.argv[1]); cmd = sys.argv[sys.argv.index("--") + 1:]
os.environ["TMPDIR"] = "/tmp"
out = collections.OrderedDict()
if os.environ.get("PMC_SETS"):
    SETS = [SETS[int(j)] for j in os.environ["PMC_SETS"].split(",")]
for i, s in enumerate(SETS):
    d = f"/tmp/pmc_{os.getpid()}_{i}"
    try:
      r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *s.split(), "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                       cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=int(os.environ.get("PMC_TIMEOUT", "180")))
    except subprocess.TimeoutExpired:
        print(f"# set {i} ({s}): timed out"); continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(f"# set {i}: no counter output (rc={r.returncode})\n" + r.stdout[-600:]); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(files[0])):
        if rx.search(row["Kernel_Name"]):
            key = (re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])[:70], row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
    for (k, c), (v, n) in acc.items():
        out.setdefault(k, []).append((c, v / n, n))
for k, rows in out.items():
    print(f"## {k}")
    for c, v, n in rows:
        print(f"   {c:32s} {v:16.0f}   (avg of {n} dispatches)")
