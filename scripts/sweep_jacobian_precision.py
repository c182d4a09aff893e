"""BASELINE config C5's "fp32-vs-fp64 Jacobian sweep": the pipeline's CAMERA_PARAMETER_RELAX schedule (pipeline.cpp:601-631:
focal length; + radial BROWN2; + BROWN24; + principal point and BROWN246) on a camera grid with tracks whose group starts with
a 3 % wrong focal length and no distortion knowledge, once with the ray blocks' Jacobians propagated in fp64 (the path) and
once in fp32 (OCHIP_TEST_HOOKS=jacobian_fp32; values, J'J accumulation and solve stay fp64).  Prints one JSON object per
precision: LM iterations, final cost, the intrinsics reached, pose error against the truth.  Run on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys
sys.path.insert(0, "tests")
import numpy as np
from opencalibration_amd import capi, host
from relax_fixtures import axis_angle, pack_edges_with_features, qangle, qmul, project, DOWN
from oracle import pyoracle   # only for pack_edges_with_features' feature bookkeeping (test helper)
rows, cols, pps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
# a survey in which the intrinsics are observable: relief of +-15 % of the flying height, cameras tilted up to 0.25 rad,
# a lens with radial distortion; 4-neighbour edges whose inliers share ground points (tracks form)
true_model = np.array([600.0, 400, 300, 0.03, -0.02, 0.004, 0, 0, 800, 600])
model = np.array([600.0, 400, 300, 0, 0, 0, 0, 0, 800, 600])
rng = np.random.default_rng(2)
n, spacing, height = rows * cols, 2.0, 10.0
ground = lambda x, y: 1.5 * np.sin(x / 3.0) * np.cos(y / 4.0) + 0.02 * x
pos = np.array([[c * spacing + rng.uniform(-0.2, 0.2), r * spacing + rng.uniform(-0.2, 0.2), height + rng.uniform(-0.5, 0.5)]
                for r in range(rows) for c in range(cols)])
ori = np.array([qmul(qmul(axis_angle([0, 0, 1], rng.normal(0, 0.3)), axis_angle(rng.normal(size=3) * [1, 1, 0], rng.uniform(0, 0.25))), DOWN)
                for _ in range(n)])
gx = np.linspace(-2 * spacing, (cols + 1) * spacing, pps)
gy = np.linspace(-2 * spacing, (rows + 1) * spacing, pps)
pts = np.array([[x + rng.uniform(-0.1, 0.1), y + rng.uniform(-0.1, 0.1), 0.0] for x in gx for y in gy])
pts[:, 2] = ground(pts[:, 0], pts[:, 1])
def observe(i):
    px = np.array([project(ori[i], pos[i], p, model) for p in pts])          # ideal pinhole
    xn = (px - model[1:3]) / model[0]
    r2 = np.sum(xn * xn, axis=1, keepdims=True)
    k1, k2, k3 = true_model[3:6]
    return (1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3) * xn * true_model[0] + true_model[1:3] + rng.normal(0, 0.2, px.shape)
px = [observe(i) for i in range(n)]
vis = [np.all((px[i] >= 0) & (px[i] < model[8:10]), axis=1) for i in range(n)]
edges = []
for r in range(rows):
    for c in range(cols):
        i = r * cols + c
        for dr, dc in ((0, 1), (1, 0), (0, -1), (-1, 0)):
            rr, cc = r + dr, c + dc
            if 0 <= rr < rows and 0 <= cc < cols:
                j = rr * cols + cc
                both = np.flatnonzero(vis[i] & vis[j])
                if len(both) >= 8:
                    edges.append(dict(src=i, dst=j, H=None, px=np.concatenate([px[i][both], px[j][both]], axis=1),
                                      match_index=np.arange(len(both)), dist=None, pid=both, point_ids=both))
q = np.array([qmul(ori[i], axis_angle(rng.normal(size=3) / 2, 0.02)) for i in range(n)])
start_model = model.copy()
start_model[0] *= 1.03
cx = np.linspace(-2 * spacing, (cols + 1) * spacing, 14)
cy = np.linspace(-2 * spacing, (rows + 1) * spacing, 14)
prev = host.Surface().set(np.zeros((0, 3)), np.zeros((0, 5), np.uint64), np.array([[x, y, ground(x, y)] for x in cx for y in cy]))
pk, feats = pack_edges_with_features(pyoracle, n, edges)
ctx = capi.Context(0)
schedule = [["FOCAL_LENGTH"], ["FOCAL_LENGTH"], ["FOCAL_LENGTH", "LENS_DISTORTIONS_RADIAL", "BROWN2"],
            ["FOCAL_LENGTH", "LENS_DISTORTIONS_RADIAL", "BROWN24"],
            ["FOCAL_LENGTH", "PRINCIPAL_POINT", "LENS_DISTORTIONS_RADIAL", "BROWN246"],
            ["FOCAL_LENGTH", "PRINCIPAL_POINT", "LENS_DISTORTIONS_RADIAL", "BROWN246"]]
cm, iters, log = start_model.copy(), 0, []
for opts in schedule:
    out = host.relax(ctx, pos, ori, model, feats, np.arange(n), q, pk, host.relax_options("ORIENTATION", "GROUND_MESH", *opts), 0.1,
                     previous=prev, cam_model=cm)
    q, cm, prev = out["orientation"], out["cam_model"], out["surface"]
    iters += int(out["iterations_total"])
    log.append(dict(options=opts, iterations=int(out["iterations_total"]), final_cost=out["final_cost"], focal=cm[0],
                    k=[cm[3], cm[4], cm[5]], pp=[cm[1], cm[2]]))
err = [qangle(q[i], ori[i]) for i in range(n)]
print(json.dumps(dict(cameras=n, residual_blocks=int(out["residual_blocks"]), unknowns=int(out["unknowns"]), lm_iterations=iters,
                      final_cost=out["final_cost"], focal=cm[0], focal_true=600.0, k=[cm[3], cm[4], cm[5]], k_true=[0.03, -0.02, 0.004],
                      pp=[cm[1], cm[2]], median_pose_error_rad=float(np.median(err)), max_pose_error_rad=float(np.max(err)), rounds=log)))
'''


def run(fp32, rows, cols, pts):
    env = dict(os.environ)
    env.pop("OCHIP_TEST_HOOKS", None)
    if fp32:
        env["OCHIP_TEST_HOOKS"] = "jacobian_fp32"
    r = subprocess.run([sys.executable, "-c", CHILD, str(rows), str(cols), str(pts)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1500)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-4000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    rows, cols, pts = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 10, 30)
    out = {"fp64": run(False, rows, cols, pts), "fp32_jacobian": run(True, rows, cols, pts)}
    a, b = out["fp64"], out["fp32_jacobian"]
    out["difference"] = dict(focal=b["focal"] - a["focal"], k=[x - y for x, y in zip(b["k"], a["k"])],
                             lm_iterations=b["lm_iterations"] - a["lm_iterations"],
                             final_cost_relative=(b["final_cost"] - a["final_cost"]) / max(abs(a["final_cost"]), 1e-300),
                             median_pose_error_rad=b["median_pose_error_rad"] - a["median_pose_error_rad"])
    print(json.dumps(out, indent=1))
