"""Print the per-kernel table of the latest rocprofv3 kernel_stats csv under gpurun_out/prof_r1b (per-image microseconds)."""
import csv, glob, sys
n_images = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
f = sorted(glob.glob('gpurun_out/prof_r1b/trace/*/*_kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 26]:
    nm = r['Name'].replace('(anonymous namespace)::', '')[:58]
    print(f"{nm:58s} calls {int(r['Calls']):5d} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} us/img {float(r['TotalDurationNs'])/1e3/n_images:7.1f} max_us {float(r['MaxNs'])/1e3:8.1f}")
