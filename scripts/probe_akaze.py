import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opencalibration_amd import capi, host, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t = time.time(); base = [synth.render_blobs(1600, 1200, s) for s in range(2)]; print("render", time.time() - t)
imgs = np.stack([base[i % 2] for i in range(B)])
ctx = capi.Context(0)
for it in range(3):
    ctx.profile_reset()
    t = time.time(); out, wh = ctx.akaze_batch(imgs, max_kp=40000); wall = time.time() - t
    n, ms = ctx.profile_get(capi.K_AKAZE)
    print(f"B={B} wall {wall*1e3:.1f} ms, device section {ms:.2f} ms = {ms/B:.2f} ms/image; keypoints {[len(o[0]) for o in out][:4]}")
t = time.time(); f = host.extract_features_batch(ctx, imgs, 40000); print("extract_features_batch", time.time() - t, [(len(x[1]), x[3]) for x in f][:2])
