"""Condense the rocprofv3 outputs of scripts/profile_r1c.sh into the small files kept under profiles/:
  <out>_kernel_stats.csv   the --stats table of the traced bench run (C3, default bench command)
  <out>_bench_line.json    the bench JSON line printed under the tracer
  <out>_pmc_hbm.json       FETCH_SIZE / WRITE_SIZE per kernel of one C2 step (200 images), with the calibration of
                           the 4-byte-per-lane access pattern against gray4_kernel's known byte counts
"""
import collections
import csv
import glob
import json
import shutil
import sys

src, out = sys.argv[1], sys.argv[2]
stats = sorted(glob.glob(src + "/trace/*/*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[-1], out + "_kernel_stats.csv")
staged = sorted(glob.glob(src + "/trace_staged/*/*_kernel_stats.csv"))
if staged:
    shutil.copy(staged[-1], out + "_staged_kernel_stats.csv")
for line in open(src + "/bench_trace.log", errors="replace"):
    if line.startswith('{"metric'):
        open(out + "_bench_line.json", "w").write(line)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


pmc = {}
for counter, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(src + "/" + d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row.get("Kernel_Name", "?"))
            acc[k][0] += 1
            acc[k][1] += float(row.get("Counter_Value", 0))
    pmc[counter] = {k: {"dispatches": v[0], "sum_KB": v[1]} for k, v in acc.items()}
n_images, src_px, dst_px = 200, 4000 * 3000, 1600 * 1200
# the bench extracts the C2 grid more than once per run (timed step, the staged step, the PCIe-inclusive pass): count the
# images from the launches of the first extract kernel, one launch per chunk of 100 images
for first in ("resize_area_lds_kernel<true>", "gray4_kernel"):
    if first in pmc["FETCH_SIZE"]:
        n_images = 100 * pmc["FETCH_SIZE"][first]["dispatches"]
        break
cal = {}
# calibration of the counters against a kernel whose HBM bytes are known: gray4_kernel (3 B read, 1 B written per source
# pixel) when the separate grey pass runs, else the fused grey + resize kernel (reads the BGR source once - its window
# overlaps are served by the L2 -, writes one float per working-image pixel)
g = pmc["FETCH_SIZE"].get("gray4_kernel")
r = pmc["FETCH_SIZE"].get("resize_area_lds_kernel<true>")
if g:
    cal["fetch_true_over_reported"] = n_images * src_px * 3 / (g["sum_KB"] * 1024.0)
elif r:
    # the fused kernel re-reads the overlap of neighbouring source windows (a few per cent), so it cannot pin the factor;
    # the guide's gfx950 value, which gray4_kernel reproduced to 5 digits in this round's earlier profiles (1.99994), stands
    cal["fetch_true_over_reported"] = 2.0
    cal["fused_resize_fetch_over_source_bytes"] = r["sum_KB"] * 1024.0 * 2.0 / (n_images * src_px * 3)
g = pmc["WRITE_SIZE"].get("gray4_kernel")
r = pmc["WRITE_SIZE"].get("resize_area_lds_kernel<true>")
if g:
    cal["write_true_over_reported"] = n_images * src_px / (g["sum_KB"] * 1024.0)
elif r:
    cal["write_true_over_reported"] = n_images * dst_px * 4 / (r["sum_KB"] * 1024.0)
extract = ["gray4_kernel", "gray_kernel", "resize_area_kernel", "resize_area_lds_kernel", "to_float_kernel", "blur_fused_kernel",
           "hmax_reduce_kernel", "hist_kernel", "kcontrast_kernel", "halfsample_kernel", "copy_plane_kernel", "nld_fused_kernel",
           "det_maxima_kernel", "scan_tiles_kernel", "collect_tiles_kernel", "suppress_kernel", "live_list_kernel", "describe_kernel",
           "rank_scan_kernel",
           "compact_ordered_kernel"]


def total(counter):
    return sum(v["sum_KB"] for k, v in pmc[counter].items() if any(k.startswith(e) for e in extract)) * 1024.0


summary = {"workload": "bench.py --config C2 --steps 1 --warmup 0 (200 images, one extract pass)", "counters": pmc,
           "calibration": dict(cal, note="against a kernel of known HBM bytes with the same 4-byte-per-lane accesses as the "
                                         "stencil kernels (gray4_kernel: 3 B read + 1 B written per source pixel; since the "
                                         "grey conversion is fused into the resize: resize_area_lds_kernel<true>, 36 MB read "
                                         "+ 7.68 MB written per image); MI355X_MICROARCH.md (HBM) says widths other than "
                                         "16 B/lane must be calibrated like this"),
           "extract_reported_bytes_per_image": {"fetch": total("FETCH_SIZE") / n_images, "write": total("WRITE_SIZE") / n_images}}
if "fetch_true_over_reported" in cal and "write_true_over_reported" in cal:
    summary["extract_hbm_bytes_per_image"] = (total("FETCH_SIZE") * cal["fetch_true_over_reported"] +
                                              total("WRITE_SIZE") * cal["write_true_over_reported"]) / n_images
json.dump(summary, open(out + "_pmc_hbm.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k != "counters"}, indent=1))
