"""profiles/r03_e2e_pmc_hbm.json from the round-3 counter passes (scripts/profile_r3_pmc.sh -> r03_extract_pmc_counters.json):
HBM bytes per image of the extract sequence - the device tail of extract_features (features.hip, std_sort.hip) included -
with the conventions of scripts/summarise_profile.py: FETCH_SIZE and WRITE_SIZE are in KB, true fetch bytes = reported x 2
on gfx950 (MI355X_MICROARCH.md; pinned on gray4_kernel in round 1), writes x 1 (checked here on the fused resize kernel).
Counter values in the input are sums over a kernel's dispatches of one C2 extraction pass."""
import json
import sys

src, out = sys.argv[1], sys.argv[2]
d = json.load(open(src))
d = d.get("kernels", d)
extract = ("gray", "resize_area", "to_float", "blur_fused", "hmax_reduce", "hist_kernel", "kcontrast", "halfsample", "copy_plane",
           "nld_fused", "det_maxima", "scan_tiles", "collect_tiles", "suppress_kernel", "live_list", "describe_kernel", "rank_scan",
           "compact_ordered", "feat_", "sort_")
r = next(v for k, v in d.items() if k.startswith("resize_area_lds_kernel"))
n_images = 100 * int(r["dispatches"])
src_px, dst_px = 4000 * 3000, 1600 * 1200
per_kernel = {}
fetch = write = 0.0
for k, v in d.items():
    if not any(k.startswith(e) for e in extract) or "FETCH_SIZE" not in v:
        continue
    per_kernel[k] = {"dispatches": v["dispatches"], "fetch_KB": v["FETCH_SIZE"], "write_KB": v["WRITE_SIZE"]}
    fetch += v["FETCH_SIZE"] * 1024.0
    write += v["WRITE_SIZE"] * 1024.0
summary = {
    "workload": "bench.py --config C2 --steps 1 --warmup 0 (one extract pass), one rocprofv3 --pmc pass per counter, no trace domain",
    "images": n_images,
    "kernels": per_kernel,
    "calibration": {"fetch_true_over_reported": 2.0,
                    "write_true_over_reported": n_images * dst_px * 4 / (r["WRITE_SIZE"] * 1024.0),
                    "fused_resize_fetch_over_source_bytes": r["FETCH_SIZE"] * 1024.0 * 2.0 / (n_images * src_px * 3)},
    "extract_reported_bytes_per_image": {"fetch": fetch / n_images, "write": write / n_images},
    "extract_hbm_bytes_per_image": (2.0 * fetch + write) / n_images,
}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("images", "calibration", "extract_hbm_bytes_per_image")}, indent=1))
