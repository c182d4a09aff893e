import sys; sys.path.insert(0, "/root/repo")
import torch, time
from opencalibration_amd import capi, host, pipeline, synth
cfg = synth.CONFIGS["C2"]
grid = synth.make_grid(seed=1, rows=cfg["rows"], cols=cfg["cols"], feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
start = pipeline.perturbed_orientations(grid, 0.1, 99)
for i in range(12):
    g, res, t = pipeline.run(ctx, grid, images, shape, start, overlap=(i % 2 == 0))
    g.close()
    free, total = torch.cuda.mem_get_info()
    import resource
    print(i, "gpu used GB %.2f" % ((total - free) / 1e9), "host rss GB %.2f" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
