"""HBM traffic of the rasterizer kernels from rocprofv3 PMC counters (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel trace
only -- MI355X_MICROARCH.md "rocprofv3 PMC slots"), with the counters calibrated in the same passes on known-size copies of
16 / 12 / 8 / 4 bytes per lane (scripts/ubench/hbm_calib.hip; the guide: FETCH_SIZE reports half the bytes of a 16 B/lane
streaming read on gfx950, other widths and WRITE_SIZE must be calibrated in the kernel's own access pattern).

usage (GPU box): python scripts/pmc_traffic.py <N gaussians> <views> <out.json> [views per batched launch set]
writes {"<N>": {stage: MB per view (corrected), ...}, "_detail": {...}}; stage names are the C-ABI entry points bench.py times."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, views, outp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
batch = sys.argv[4] if len(sys.argv) > 4 else "1"
os.environ["TMPDIR"] = "/tmp"
STAGE = [  # kernel-name regex -> bench.py stage
    (r"k_project_sh_fwd", "gc_project_sh_fwd"), (r"k_project_sh_bwd", "gc_project_sh_bwd"),
    (r"k_rasterize_fwd", "gc_rasterize_fwd"), (r"k_rasterize_bwd", "gc_rasterize_bwd"),
    (r"k_depth_keys|k_radix_hist<true>|k_radix_scatter<true>|k_gather_tiles|k_scan_|k_table_scan|k_tri_hist|k_tri_scatter|k_box_counts", "gc_raster_depth_order"),
    (r"k_radix_hist<false>|k_radix_scatter|k_emit_sorted|k_tile_bins", "gc_raster_bin_tiles_dev"),
    (r"k_ssim|k_raster_finalize", "loss+finalize"),
    (r"k_calib_copy<.*4u>", "calib16"), (r"k_calib_copy<.*f3>", "calib12"),
    (r"k_calib_copy<.*2u>", "calib8"), (r"k_calib_copy<float>", "calib4"),
]
raw = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = f"/tmp/pmct_{os.getpid()}_{counter}"
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                        sys.executable, os.path.join(ROOT, "scripts", "raster_traffic_target.py"), str(N), str(views), batch],
                       cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(f"# {counter}: no counter output (rc={r.returncode})\n" + r.stdout[-1500:]); continue
    m = re.search(r"calib_bytes (\d+)", r.stdout)
    calib_bytes = int(m.group(1)) if m else 0
    acc = collections.defaultdict(lambda: [0.0, 0])
    names = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        kn = row["Kernel_Name"]
        for rx, st in STAGE:
            if re.search(rx, kn):
                acc[st][0] += float(row["Counter_Value"]); acc[st][1] += 1
                names[(st, re.sub(r"\(anonymous namespace\)::", "", kn)[:60])] += 1
                break
    raw[counter] = {"calib_bytes": calib_bytes, "sum": {k: v[0] for k, v in acc.items()}, "dispatches": {k: v[1] for k, v in acc.items()},
                    "kernels": {f"{a}: {b}": c for (a, b), c in names.items()}, "stdout_tail": r.stdout[-300:]}
# per-view corrected traffic: each counter is scaled by (known bytes / counted bytes) of the 16 B/lane calibration copy of the same pass
# (gfx950: FETCH_SIZE counts half the bytes of every width tried, WRITE_SIZE is exact; the factors are re-measured here, not assumed)
per_view = {}
corr = {}
for counter, r in raw.items():
    cal = r["sum"].get("calib16", 0.0) * 1024.0
    corr[counter] = (2.0 * r["calib_bytes"] / cal) if cal else None          # two calibration dispatches of calib_bytes each
    for st, v in r["sum"].items():
        if st.startswith("calib") or corr[counter] is None:
            continue
        per_view.setdefault(st, {})[counter] = v * 1024.0 * corr[counter] / views / 1e6
table = {st: round(sum(c.values()), 1) for st, c in per_view.items() if len(c) == 2}
out = {str(N): table, "_per_counter_MB_per_view": {st: {k: round(x, 1) for k, x in c.items()} for st, c in per_view.items()},
       "_correction": corr, "_views": views, "_views_per_launch_set": int(batch), "_detail": raw,
       "_note": "MB per view = FETCH_SIZE + WRITE_SIZE (KB as reported by rocprofv3, separate passes), each scaled by the calibration "
                "factor measured in the same pass on a 768 MiB copy (see DESIGN.md 4)"}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
