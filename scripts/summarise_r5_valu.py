"""profiles/r05_extract_valu.json from a counter pass over scripts/extract_only.py (100 views of the C2 grid, extracted once):
vector instructions issued per image by every kernel of the extract sequence, against the part's issue peak.  A wavefront
instruction occupies its SIMD for 4 cycles (64 lanes over 16): 256 CUs x 4 SIMDs x 2.4 GHz / 4 = 614.4 G wavefront
instructions per second at the top clock.  SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE give the clock the kernels actually ran at."""
import json
import sys

src, out, stats_csv = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else None)
d = json.load(open(src))
d = d.get("kernels", d)
skip = ("render_views", "__amd")
r = next(v for k, v in d.items() if k.startswith("resize_area_lds_kernel"))
n_images = 100 * int(r["dispatches"])
PEAK = 256 * 4 * 2.4e9 / 4
us = {}
if stats_csv:
    import csv
    for row in csv.DictReader(open(stats_csv)):
        name = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        us[name] = us.get(name, 0.0) + float(row["TotalDurationNs"])
    traced_images = None
per_kernel, total = {}, 0.0
for k, v in d.items():
    if any(k.startswith(e) for e in skip) or "SQ_INSTS_VALU" not in v:
        continue
    w = v["SQ_INSTS_VALU"] / n_images
    per_kernel[k] = {"dispatches": v["dispatches"], "valu_wave_instructions_per_image": round(w, 1),
                     "issue_us_per_image_at_2.4GHz": round(w / PEAK * 1e6, 3)}
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
        per_kernel[k]["valu_active_share_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 4)
    total += w
summary = {
    "workload": "scripts/extract_only.py 100 1 (100 views of the C2 grid, one extract pass), rocprofv3 --pmc passes, no trace domain",
    "images": n_images,
    "issue_peak_wave_instructions_per_s": PEAK,
    "extract_valu_wave_instructions_per_image": round(total, 1),
    "extract_issue_us_per_image_at_2.4GHz": round(total / PEAK * 1e6, 2),
    "kernels": dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]["valu_wave_instructions_per_image"])),
}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("images", "extract_valu_wave_instructions_per_image", "extract_issue_us_per_image_at_2.4GHz")}, indent=1))
for k, v in list(summary["kernels"].items())[:14]:
    print("  %-44s %12.0f  %7.2f us" % (k[:44], v["valu_wave_instructions_per_image"], v["issue_us_per_image_at_2.4GHz"]))
