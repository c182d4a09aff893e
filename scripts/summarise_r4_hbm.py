"""profiles/r04_e2e_pmc_hbm.json from the round-4 FETCH_SIZE / WRITE_SIZE passes over scripts/extract_only.py (100 views of the
C2 grid, extracted once): HBM bytes per image of the extract sequence, device tail included.  Conventions of
MI355X_MICROARCH.md / scripts/summarise_profile.py: FETCH_SIZE and WRITE_SIZE are in KB; true fetch bytes = reported x 2 on
gfx950 (pinned on gray4_kernel in round 1), writes x 1 (checked here on the fused resize kernel's known output)."""
import json
import sys

src, out = sys.argv[1], sys.argv[2]
d = json.load(open(src))
d = d.get("kernels", d)
skip = ("render_views", "__amd")
r = next(v for k, v in d.items() if k.startswith("resize_area_lds_kernel"))
n_images = 100 * int(r["dispatches"])
src_px, dst_px = 4000 * 3000, 1600 * 1200
per_kernel, fetch, write = {}, 0.0, 0.0
for k, v in d.items():
    if any(k.startswith(e) for e in skip) or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    per_kernel[k] = {"dispatches": v["dispatches"], "fetch_KB": v["FETCH_SIZE"], "write_KB": v["WRITE_SIZE"],
                     "hbm_MB_per_image": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / n_images / 1e6}
    fetch += v["FETCH_SIZE"] * 1024.0
    write += v["WRITE_SIZE"] * 1024.0
summary = {
    "workload": "scripts/extract_only.py 100 1 (100 views of the C2 grid, one extract pass), one rocprofv3 --pmc pass per counter, no trace domain",
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
