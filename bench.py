#!/usr/bin/env python3
"""bench.py — images/sec end-to-end (extract + match + relax) of the opencalibration hot path on MI355X
(BASELINE.json metric), plus LM iters/sec.

One "step" = one pass of the hot path over one synthetic aerial grid (default C3: 1 000 images, 25x40 grid):
  extract   AKAZE features of every 4000x3000 view (views are rendered into HBM before the timed region)
  link      kNN(10) pairs: 40 px subsample -> Hamming 2-NN -> ratio/sort -> homography RANSAC -> decompose
  relax     ground-plane bundle adjustment of all cameras as one group (Levenberg-Marquardt)
N > 1: one process per GPU (torch.distributed over RCCL).  `python bench.py --gpus N` starts the N ranks itself when no
launcher did (under `python -m torch.distributed.run` its RANK / WORLD_SIZE are honoured).
  --scaling weak    (default) every rank runs its own grid of the same shape (surveys are independent units: no data-path
                    collective), value = total images / max time; the line also carries a short run of the strong mode
  --scaling strong  ONE survey over the N ranks (BASELINE config C4): images extracted by contiguous block, directed pairs
                    linked by the rank that owns them, the 40 px subsets and the pairs' results all-gathered (RCCL); the
                    relax either pipelined over surveys (default: rank k mod N relaxes survey k in the shadow of the next
                    surveys, as at N = 1) or sharded inside the step (--relax sharded: residual blocks over all ranks,
                    one RCCL exchange per evaluation)

Prints ONE JSON line on rank 0.  `roofline` describes the dominant device kernel, timed with HIP events on
the stream it is launched on; `cpu_baseline` is the CPU restatement (oracle/) timed on this box's usable host
cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The share of an 8-rank node: OCHIP_BENCH_CPUS=2 pins the process to that many CPUs before any thread pool exists.
if os.environ.get("OCHIP_BENCH_CPUS"):
    _cpus = sorted(os.sched_getaffinity(0))[:max(1, int(os.environ["OCHIP_BENCH_CPUS"]))]
    os.sched_setaffinity(0, _cpus)
# The step keeps ~8 HIP streams busy (4 extraction sequences, 3 link runners, the relax of the previous survey, copies).
# The ROCm runtime maps streams onto 4 hardware queues by default, so a short latency-bound launch can sit behind another
# stream's long kernel in the same queue; 16 queues measured +5-6 % images/s.  Read by the runtime
# when it initialises, hence set before anything touches HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# RCCL across processes needs dmabuf IPC on this driver (already exported on the pool's boxes; kept if it is not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# A survey's feature lists (1.8 MB per image) are allocated by the host tail and freed with the graph one step later.
# glibc hands such blocks straight back to the kernel (mmap threshold, heap trimming), so every step would page-fault its
# 2 GB in again, 4 KB at a time, inside the tail's OpenMP team; a long-running pipeline process keeps them.
import ctypes as _ctypes

_libc = _ctypes.CDLL("libc.so.6")
_libc.mallopt(-3, 32 << 20)   # M_MMAP_THRESHOLD: its maximum
_libc.mallopt(-1, 1 << 30)    # M_TRIM_THRESHOLD
_libc.mallopt(-2, 64 << 20)   # M_TOP_PAD

METRIC = "images/sec end-to-end (extract+match+relax) on synthetic aerial grid; LM iters/sec"
DTYPE = "f32 (extract) + fp4 0/1 bits, f32 accumulate, exact (match) + f64 (RANSAC, relax)"


def _env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def fed_steps_per_level():
    """FED steps of the 16 levels (4 octaves x 4 sublevels, sigma0 1.6, tau_max 0.25): AKAZE's fed_tau_by_process_time."""
    sig = [1.6 * 2.0 ** (j / 4.0 + o) for o in range(4) for j in range(4)]
    return [0] + [int(np.ceil(np.sqrt(3.0 * (0.5 * (sig[i] ** 2 - sig[i - 1] ** 2)) / 0.25 + 0.25) - 0.5 - 1e-8))
                  for i in range(1, 16)]


def extract_algorithmic_bytes(w, h, reference_passes=False):
    """Algorithmic HBM bytes of the extract (AKAZE) launch sequence per image: every data-dependent pass of the CURRENT pass
    structure reads its inputs and writes its outputs once; what a launch keeps in registers is not
    charged.  Round 5 (level_strip_kernel: Lsmooth, conductivity, (Lx, Ly) and the first group of <= 4 FED steps in one
    launch; grey conversion inside the resize):
      source: BGR read 3 B / full-resolution px; working image write 1 float / working px
      k-contrast: image read 1, gradient magnitude write 1 + read 1
      level 0: Gaussian(1.6) read 1 + write 1 and, in the same launch, its derivatives write 2 (Lx, Ly); determinant read 2
      level i >= 1: level launch read L 1, write (Lx, Ly) 2, write L 1 (+ conductivity write 1 when FED groups follow);
        every further group of <= 4 FED steps read L 1 + conductivity 1, write L 1; determinant read 2
      description: L, Lx, Ly read once by the sampler 3 per level (the maxima maps are sparse: not charged)
    reference_passes=True: the pass structure of rounds 2 - 4 (Lsmooth pass, every FED group and the determinant as launches
    of their own, grey written and read): 0.756 GB per 4000 x 3000 image, kept for the comparison across rounds."""
    sc = min(1.0, 1600.0 / max(w, h))
    W, H = int(round(w * sc)), int(round(h * sc))
    fed = fed_steps_per_level()
    if reference_passes:
        px_floats = 5.0 * W * H
        for lvl in range(16):
            px = (W >> (lvl // 4)) * (H >> (lvl // 4))
            px_floats += px * (9 + (4 + 3 * ((fed[lvl] + 3) // 4) if lvl else 3))
        return 4.0 * px_floats + w * h * 5.0
    px_floats = 1.0 * W * H + 3.0 * W * H      # working image; k-contrast
    for lvl in range(16):
        px = (W >> (lvl // 4)) * (H >> (lvl // 4))
        if lvl == 0:
            per = 4 + 2 + 3
        else:
            groups = (fed[lvl] + 3) // 4
            per = 4 + (1 if groups > 1 else 0) + 3 * max(groups - 1, 0) + 2 + 3
        px_floats += px * per
    return 4.0 * px_floats + w * h * 3.0


def committed_valu_per_image():
    """Vector (VALU) wavefront instructions per image of the extract sequence from the committed counter pass
    (profiles/r06_extract_valu.json, else round 5's): (instructions, peak per second, file) or (None, None, None)."""
    for name in ("r06_extract_valu.json", "r05_extract_valu.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                d = json.load(fh)
            return float(d["extract_valu_wave_instructions_per_image"]), float(d["issue_peak_wave_instructions_per_s"]), "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def committed_traffic_per_image():
    """HBM bytes per image of the extract sequence from the committed PMC pass (FETCH_SIZE x 2 + WRITE_SIZE per the guide's
    gfx950 correction, scripts/summarise_profile.py): (bytes, file) or (None, None).  The counters cannot be read from
    inside this process; the figure is a constant of the code version the file was taken with."""
    for name in ("r06_e2e_pmc_hbm.json", "r05_e2e_pmc_hbm.json", "r04_e2e_pmc_hbm.json", "r03_e2e_pmc_hbm.json", "r02_e2e_pmc_hbm.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                return float(json.load(fh)["extract_hbm_bytes_per_image"]), "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, exactly as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    This process has not touched HIP (nothing but the standard library and numpy is imported yet), it only waits for its
    children, so no process that has initialised a GPU ever starts another program."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        # a rank that dies leaves the others waiting in a collective: once one has failed the rest are stopped
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


class Proc:
    """This rank: its place in the job, its device, barrier and reductions."""

    def __init__(self, args):
        self.rank, self.world, self.local_rank = _env_int("RANK", 0), _env_int("WORLD_SIZE", 1), _env_int("LOCAL_RANK", 0)
        from opencalibration_amd import host as _host_mod

        self.cores = _host_mod.effective_cpus()   # affinity mask capped by the cgroup CPU quota
        self.threads = max(1, self.cores // max(self.world, 1))                       # CPU baseline: one thread per usable core
        omp_threads = max(1, _host_mod.host_threads() // max(self.world, 1))           # host phases of the hot path (bursty, see host.py)
        # torch.distributed.run exports OMP_NUM_THREADS=1 for every worker unless the user set it; that default is not a
        # choice made for this program (its host phases are OpenMP-parallel), so it is replaced.  OCHIP_HOST_THREADS pins it.
        os.environ["OMP_NUM_THREADS"] = str(int(os.environ.get("OCHIP_HOST_THREADS", omp_threads)))
        # idle team members sleep instead of spinning, and host threads waiting for the device block instead of polling: on a
        # box whose CPU quota is smaller than the thread count the spinning was throttling the working threads (C3, 16-CPU
        # quota: 668 -> 591 ms per step, 30 % less CPU time)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("OCHIP_BLOCKING_SYNC", "1")

        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.preflight_s = None
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
        # one rank per GPU over RCCL ("nccl").  OCHIP_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 code path
        # (barriers, thread split, exchanges, max-over-ranks) run on a box with fewer GPUs than ranks, ranks sharing devices.
        self.backend = os.environ.get("OCHIP_BENCH_BACKEND", "nccl")
        self.device_index = self.local_rank if self.backend == "nccl" else self.local_rank % torch.cuda.device_count()
        torch.cuda.set_device(self.device_index)
        if self.world > 1:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device_index))
            else:
                dist.init_process_group(self.backend)
            self.world = dist.get_world_size()       # the ranks that really joined the communicator
            self.preflight_s = self._preflight()

    def _preflight(self):
        """The first collective of a scaling run is also the first time these ranks talk: a 4 MB all-gather (the
        size of the relax exchange) checked against what every rank must hold afterwards, under a watchdog, so
        that a transport that cannot work ends the job with one line and a non-zero code instead of hanging in the first
        exchange of the timed region."""
        torch, dist = self.torch, self.dist
        limit = 60.0
        done = threading.Event()

        def watchdog():
            if not done.wait(limit):
                sys.stderr.write(f"bench.py: rank {self.rank}: the {self.backend} pre-flight all-gather did not finish in "
                                 f"{limit:.0f} s (communicator of {self.world} ranks); giving up\n")
                sys.stderr.flush()
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        t0 = time.perf_counter()
        n = (4 << 20) // 8 // self.world
        dev = "cuda" if self.backend == "nccl" else "cpu"
        full = torch.zeros(self.world * n, dtype=torch.float64, device=dev)
        mine = torch.arange(n, dtype=torch.float64, device=dev) + float(self.rank) * 1e6
        dist.all_gather_into_tensor(full, mine)
        if dev == "cuda":
            torch.cuda.synchronize()
        expect = (torch.arange(n, dtype=torch.float64).repeat(self.world)
                  + torch.arange(self.world, dtype=torch.float64).repeat_interleave(n) * 1e6)
        ok = bool(torch.equal(full.cpu(), expect))
        done.set()
        if not ok:
            sys.stderr.write(f"bench.py: rank {self.rank}: the {self.backend} pre-flight all-gather returned wrong data\n")
            sys.stderr.flush()
            os._exit(4)
        return round(time.perf_counter() - t0, 3)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        tt = self.torch.tensor([value], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        if self.world > 1:
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def gather_objects(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def finish(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def relax_memory_report(ctx):
    """The largest reduced system the relaxes of this process held: unknowns, bytes of J'J + factor as stored (tiles of the
    block envelope) and what the two matrices would take dense."""
    n, stored, dense = ctx.relax_memory()
    return {"unknowns": n, "stored_mbytes": round(stored / 1e6, 2), "dense_mbytes": round(dense / 1e6, 2)}


def valu_issue(images, seconds):
    """What bounds the extract kernels since round 5 is the rate at which a CU issues vector instructions (the counters have
    every large kernel of the sequence issue-bound: DESIGN.md section 4.1), so the sequence is also priced against that:
    wavefront instructions per image from the committed counter pass x images / seconds, against 256 CUs x 4 SIMDs x
    2.4 GHz / 4 cycles per wavefront instruction."""
    n, peak, src = committed_valu_per_image()
    if n is None:
        return None
    ach = n * images / seconds
    return {"bound": "vector issue", "achieved": round(ach / 1e9, 1), "peak": round(peak / 1e9, 1), "unit": "1e9 wavefront instructions/s",
            "frac": round(ach / peak, 4), "wave_instructions_per_image": round(n),
            "source": f"{src} (rocprofv3 --pmc SQ_INSTS_VALU over the extract stage alone; not re-measured in this run); the peak is at "
                      "the 2.4 GHz top clock - under these kernels the part runs nearer 1.7 - 2.0 GHz (SQ_BUSY_CYCLES in the same file)"}


def device_rooflines(ctx, capi, steps, images_per_step, t_extract_per_step, shape, overlap, link_work, edges, staged=None):
    """The `roofline` object: the extract launch sequence against HBM, with the match (VALU) and relax (MFMA) entries."""
    def prof(kid):
        n, ms = ctx.profile_get(kid)
        return int(n), float(ms)

    n_match, ms_match = prof(capi.K_MATCH)
    n_ransac, ms_ransac = prof(capi.K_RANSAC)
    n_eval, ms_eval = prof(capi.K_RELAX_EVAL)
    n_solve, ms_solve = prof(capi.K_RELAX_SOLVE)
    n_akaze, ms_akaze = prof(capi.K_AKAZE)
    match_computed, match_delivered = ctx.match_work()
    relax_flops = ctx.relax_work()
    work = ctx.work_counters()
    _, h, w = shape
    alg_bytes_img = extract_algorithmic_bytes(w, h)
    imgs_per_launch = images_per_step * steps / max(n_akaze, 1)
    avg_ms_akaze = ms_akaze / max(n_akaze, 1)          # HIP events around one chunk's launch sequence, on its stream
    # four device contexts keep four chunks in flight (their sequences overlap in time), so the rate is taken over the
    # extract stage's wall time - device-bound, the host tail runs underneath it - which can only understate it
    achieved = alg_bytes_img * images_per_step / t_extract_per_step / 1e9
    launch_ms = t_extract_per_step * 1e3 / max(images_per_step / max(imgs_per_launch, 1e-9), 1e-9)
    traffic_img, traffic_file = committed_traffic_per_image()
    traffic = None if traffic_img is None else round(traffic_img * imgs_per_launch)
    return {
        "kernel": "extract (AKAZE) kernel sequence, one batched launch sequence per %d images, up to 4 sequences in flight"
                  "%s" % (round(imgs_per_launch), "; in the timed steps it shares the device with the link kernels" if overlap else ""),
        "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
        "frac": round(achieved / 8000.0, 4), "traffic": traffic,
        "peak_note": "8 TB/s is the part's HBM peak (MI355X_MICROARCH.md); pure streaming kernels measured on this part "
                     "(scripts/ubench_hbm.hip, profiles/r04_hbm_stream_ubench.txt) reach 6.2-6.4 TB/s read-only and 4.9-5.5 TB/s "
                     "at the read : write mixes of this sequence's passes",
        "traffic_source": None if traffic is None else f"rocprofv3 --pmc pass committed as {traffic_file} (FETCH_SIZE x 2 + "
                                                          "WRITE_SIZE per image x images per launch); not re-measured in this run",
        "avg_launch_ms": round(launch_ms, 3), "launches": n_akaze,
        "hip_event_ms_per_sequence_overlapped": round(avg_ms_akaze, 3),
        "algorithmic_bytes_per_launch": round(alg_bytes_img * imgs_per_launch),
        "algorithmic_bytes_per_image": round(alg_bytes_img),
        "algorithmic_bytes_note": "the CURRENT pass structure (round 5: a level's Lsmooth, conductivity, derivatives and first FED "
                                  "group are one launch, the grey image is never written); with the pass structure of rounds 2 - 4 "
                                  "the same launch sequence would be charged %.3f GB per image and `frac` would read %.3f"
                                  % (extract_algorithmic_bytes(w, h, True) / 1e9,
                                     extract_algorithmic_bytes(w, h, True) * images_per_step / t_extract_per_step / 8e12),
        "valu_issue": valu_issue(images_per_step, t_extract_per_step),
        "staged": staged,
        # match: the 2-NN runs on the matrix cores (hamming_2nn_mfma_kernel: popcount(a ^ b) = |a| + |b| - 2 a.b with the bits
        # as FP4 0 / 1, v_mfma_scale_f32_32x32x64_f8f6f4, 8 instructions per 32 x 32 tile of distances): 2 x 512 flop per
        # distance against the dense FP4 peak of 10 PFLOP/s = 9.8e12 distances/s.  Every direction of a pair is a job of
        # its own (the vector-pipe kernels of rounds 1-3 took both directions from one pass; they remain for images with
        # more than 8 192 features in the matcher).  HBM traffic is negligible (256 B per feature and tile pass, from L2).
        "match": {"kernel": "hamming_2nn_mfma_kernel (+ expand_fp4_kernel)", "bound": "mfma",
                  "achieved": round(match_computed / max(ms_match, 1e-9) * 1e3 / 1e12, 4), "peak": 9.8,
                  "unit": "1e12 descriptor distances/s (FP4 MFMA: 1 024 flop each, dense peak 10 PFLOP/s)",
                  "frac": round(match_computed / max(ms_match, 1e-9) * 1e3 / 9.8e12, 4),
                  "tflops": round(match_computed * 1024 / max(ms_match, 1e-9) * 1e3 / 1e12, 1),
                  "distances_per_step": round(match_computed / max(steps, 1)),
                  "features_entering_matcher_per_image": None if not link_work else round(link_work["subset_features"] / max(link_work.get("images", 1), 1), 1),
                  "directed_pairs": int(edges), "launches": n_match,
                  "device_ms_per_step": round(ms_match / max(steps, 1), 3)},
        # RANSAC scoring: fp64 on the vector pipe.  Per (hypothesis, correspondence): two 3 x 3 matrix-vector products, two
        # perspective divisions, the symmetric transfer error with its square root and the MSAC term, ~50 flops
        # (homography_model.cpp:89-118).  The count is loop trips x correspondences, an upper bound (SPRT leaves early).
        "ransac": {"kernel": "ransac_homography_kernel", "bound": "fp64 valu",
                   "achieved": round(work["ransac_hyp_corr"] * 50 / max(ms_ransac, 1e-9) * 1e3 / 1e12, 4), "peak": 78.6, "unit": "TFLOP/s",
                   "frac": round(work["ransac_hyp_corr"] * 50 / max(ms_ransac, 1e-9) * 1e3 / 78.6e12, 5),
                   "hypothesis_correspondence_pairs_per_step": round(work["ransac_hyp_corr"] / max(steps, 1)),
                   "note": "latency-bound by design: one wavefront per image pair walks the reference's sequential loop (sample, "
                           "9 x 9 LU, score with early exit, local optimisation); the scoring is a fraction of its instructions",
                   "device_ms_per_step": round(ms_ransac / max(steps, 1), 3)},
        # relax evaluation (ground-plane engine): per 2-ray residual block 56 B in (two rays, two camera indices) and its
        # share of the pair's packed J'J record out; ~4 kflop with 11-wide duals, ~0.4 kflop for the cost alone
        "relax_eval": {"kernel": "relax_pair_eval_kernel<true|false>", "bound": "neither (register-limited occupancy)",
                       "blocks_with_jacobian": work["relax_blocks_jac"], "blocks_cost_only": work["relax_blocks_cost"],
                       "achieved_tflops": round((work["relax_blocks_jac"] * 4000 + work["relax_blocks_cost"] * 400) / max(ms_eval, 1e-9) * 1e3 / 1e12, 3),
                       "frac_of_fp64_vector_peak": round((work["relax_blocks_jac"] * 4000 + work["relax_blocks_cost"] * 400) / max(ms_eval, 1e-9) * 1e3 / 78.6e12, 4),
                       "achieved_gbs": round((work["relax_blocks_jac"] + work["relax_blocks_cost"]) * 56 / max(ms_eval, 1e-9) * 1e3 / 1e9, 1),
                       "frac_of_hbm_peak": round((work["relax_blocks_jac"] + work["relax_blocks_cost"]) * 56 / max(ms_eval, 1e-9) * 1e3 / 8e12, 5),
                       "device_ms_per_step": round(ms_eval / max(steps, 1), 3)},
        # relax linear solve: the only MFMA use on the path (v_mfma_f64_16x16x4f64 in the tile products of the block-envelope
        # Cholesky, one launch per factorisation).  The factorisation is a dependency chain of 64 x 64 tiles and is bound by
        # the diagonal tiles' latency, not by the matrix cores; the dense figure is the peak the guide's FP64-matrix rate
        # gives, not a target for this path.
        "relax_mfma": {"kernel": "chol_tiles_kernel", "bound": "mfma",
                       "achieved": round(relax_flops / max(ms_solve, 1e-9) * 1e3 / 1e12, 4), "peak": 78.6, "unit": "TFLOP/s",
                       "frac": round(relax_flops / max(ms_solve, 1e-9) * 1e3 / 78.6e12, 5),
                       "note": "latency-bound: one launch per factorisation, a dependency chain of 64 x 64 tiles inside the block "
                               "envelope (critical path = the diagonal tiles); the time base is the whole linear solve (build, "
                               "factorisation, substitutions, step)",
                       "flops_per_step": round(relax_flops / max(steps, 1)), "solves": n_solve,
                       "system_memory": relax_memory_report(ctx)},
        "other_kernels_avg_ms": {
            "match launch (all pairs of a link range)": round(ms_match / max(n_match, 1), 3),
            "ransac_homography_kernel": round(ms_ransac / max(n_ransac, 1), 3),
            "relax_pair_eval_kernel": round(ms_eval / max(n_eval, 1), 4),
            "relax_linear_solve (build + tile Cholesky + substitutions + step)": round(ms_solve / max(n_solve, 1), 3)},
    }


def cpu_baseline_leg(ctx, grid, images, shape, threads, start_ori):
    """The restatement (oracle/) on a bounded sample of the same workload, all usable host cores (rank 0, N = 1 only)."""
    from concurrent.futures import ThreadPoolExecutor

    from opencalibration_amd import host, pipeline
    from oracle import pyoracle

    _, h, w = shape
    pyoracle.lib()
    # >= 64 views and >= 256 directed pairs, so the sample's own noise stays below the quoted digits
    n_ex = min(shape[0], max(64, threads))
    views = [ctx.synth_views_read(images, i, w, h) for i in range(n_ex)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:        # ctypes releases the GIL: one image per core, as the load stage does
        list(ex.map(pyoracle.extract_features, views))
    wall_extract = time.perf_counter() - t0
    t_extract = wall_extract / n_ex
    del views
    # link: first pairs of the same grid, features from the (bit-identical) device extraction
    feats = host.extract_features_batch(ctx, images, 30000, device_shape=(min(shape[0], 60), h, w))
    xy = grid.position[:len(feats), :2]
    d2 = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    knn = np.argsort(d2, axis=1, kind="stable")[:, :10]
    pairs = [(a, int(b)) for a in range(len(feats)) for b in knn[a] if b != a][:max(256, 2 * threads)]
    off = np.concatenate([[0], np.cumsum([len(f[1]) for f in feats])]).astype(np.uint64)
    loc = np.ascontiguousarray(np.concatenate([f[0] for f in feats]))
    st = np.ascontiguousarray(np.concatenate([f[1] for f in feats]))
    de = np.ascontiguousarray(np.concatenate([f[2] for f in feats]))
    ns = np.array([f[3] for f in feats], np.uint64)
    sample = np.ascontiguousarray(np.array(pairs, np.uint32))
    counts, Hs, secs = np.zeros((len(pairs), 2), np.uint64), np.zeros((len(pairs), 9)), np.zeros(4)
    pyoracle.lib().oc_link_batch_cpu(loc, st, de, off, len(feats), ns, grid.model, sample, len(pairs), 1, threads,
                                     counts, Hs, secs)
    t_link = secs[0] / (len(pairs) / 9.0)          # seconds per source image (9 directed pairs each)
    cpu = {"value": round(1.0 / (t_extract + t_link), 3), "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"extract: {n_ex} views, one per core at a time, {wall_extract:.2f} s wall; link: first {len(pairs)} "
                     f"directed pairs, OpenMP dynamic,1, faithful variant (link_stage.cpp:80-81), {secs[0]:.2f} s wall "
                     f"(cpu-seconds match/undistort/ransac {secs[1]:.1f}/{secs[2]:.2f}/{secs[3]:.1f}); value = "
                     f"1 / (extract + link seconds per image); relax timed separately below",
           "extract_s_per_image_per_core": round(t_extract * min(n_ex, threads), 3)}
    try:
        gg, _, _ = pipeline.run(ctx, grid, images, shape, start_ori)
        sub = np.arange(min(50, grid.n_images))
        e50 = gg.edges_flat(sub)
        t0 = time.perf_counter()
        r50 = pyoracle.relax_ground_plane(grid.position[sub], start_ori[sub], grid.model, sub, start_ori[sub], e50)
        tcpu = time.perf_counter() - t0
        cpu["relax_cpu"] = {"cameras": int(len(sub)), "residual_blocks": int(r50["residual_blocks"]),
                            "lm_iterations": int(r50["iterations_total"]), "seconds": round(tcpu, 3),
                            "lm_iters_per_s": round(r50["iterations_total"] / tcpu, 2), "cores": 1,
                            "note": "one reference-sized relax group (50 cameras, relax_stage.cpp:52), single "
                                    "thread like Ceres num_threads=1 (relax_problem.cpp:30)"}
        gg.close()
    except Exception as ex:  # informational only
        cpu["relax_cpu"] = {"error": str(ex)}
    return cpu


def synth_subgrid(grid, n):
    """The first n cameras of a synthetic grid as a grid of their own (poses and camera model; no features)."""
    import copy

    sub = copy.copy(grid)
    sub.n_images = n
    sub.position = grid.position[:n]
    sub.orientation = grid.orientation[:n]
    return sub


def beside_the_headline(ctx, grid, images, shape, start_ori, step_s, rctx=None):
    """Rank 0, N = 1, after the timed region; none of it is part of `value`."""
    from opencalibration_amd import host, pipeline

    n, h, w = shape
    extras = {}
    try:
        # (a) images that start in HOST memory, as the reference's boundary hands them over (cv::Mat): views copied back
        # from HBM into page-locked memory, then a whole step (load + link overlapped, relax) run from there
        # (the whole survey when its views fit 48 GB of page-locked memory - C3: 36 GB -, else its first cameras)
        n_h = min(grid.n_images, max(100, int(48e9 // (h * w * 3))))
        hostviews, release = ctx.host_array((n_h, h, w, 3))
        for i in range(n_h):
            ctx.synth_views_read_into(images, i, w, h, hostviews[i])

        def load_seconds(src, device_shape):   # the load stage alone (extract + one node per image)
            gl = host.Graph()
            ml = gl.add_model(grid.model)
            t0 = time.perf_counter()
            gl.load_images(ctx, src, ml, grid.position[:n_h], 30000, device_shape=device_shape)
            dt = time.perf_counter() - t0
            gl.close()
            return dt / n_h

        load_seconds(hostviews, None)          # (warm-up: the chunk-of-25 result buffers are allocated on first use)
        t_host = load_seconds(hostviews, None)
        t_dev = load_seconds(images, (n_h, h, w))
        # a whole step from host memory, measured: the first n_h cameras of the survey as a survey of their own (load with
        # the upload inside it, link, relax), three times
        sub = synth_subgrid(grid, n_h)
        sub_ori = start_ori[:n_h]
        pipeline.run(ctx, sub, None, (n_h, h, w), sub_ori, host_images=hostviews)[0].close()       # warm-up
        # (the relax of survey k under the load + link of survey k + 1 on the relax context, exactly as the timed steps)
        pending, ts, n_e2e = None, {}, 5

        def relax_of(g, res, t):
            pipeline.relax_step(rctx, g, sub_ori, res, t)
            g.close()

        t0 = time.perf_counter()
        for _ in range(n_e2e):
            gs, rs, ts = pipeline.run(ctx, sub, None, (n_h, h, w), sub_ori, host_images=hostviews, relax=rctx is None)
            if rctx is None:
                gs.close()
                continue
            if pending is not None:
                pending.join()
            pending = threading.Thread(target=relax_of, args=(gs, rs, ts))
            pending.start()
        if pending is not None:
            pending.join()
        e2e = n_e2e * n_h / (time.perf_counter() - t0)
        # Two surveys in flight (each on its own set of device contexts): from host memory a step is bound by the PCIe link,
        # not by the device, so the link tail of one survey and the ramps of its first / last chunks can hide under the
        # uploads of the next - the reference's pipeline keeps consecutive batches in flight the same way
        # (pipeline.cpp:543-560).  Relaxes one at a time on the relax context.
        e2e_two = None
        if rctx is not None:
            from opencalibration_amd import capi as _capi

            ctx_b = _capi.Context(getattr(ctx, "device", 0))
            relax_lock, relax_threads, errors = threading.Lock(), [], []

            def relax_locked(g, res, t):
                with relax_lock:
                    pipeline.relax_step(rctx, g, sub_ori, res, t)
                g.close()

            def lane(c, steps):
                try:
                    for _ in range(steps):
                        gs, rs, ts2 = pipeline.run(c, sub, None, (n_h, h, w), sub_ori, host_images=hostviews, relax=False)
                        th = threading.Thread(target=relax_locked, args=(gs, rs, ts2))
                        th.start()
                        relax_threads.append(th)
                except Exception as ex:
                    errors.append(ex)

            pipeline.run(ctx_b, sub, None, (n_h, h, w), sub_ori, host_images=hostviews, relax=False)[0].close()    # warm-up
            per_lane = 4
            t0 = time.perf_counter()
            lanes = [threading.Thread(target=lane, args=(c, per_lane)) for c in (ctx, ctx_b)]
            lanes[0].start()
            time.sleep(0.3)                     # (stagger: one survey's link tail under the other's uploads)
            lanes[1].start()
            for th in lanes:
                th.join()
            for th in relax_threads:
                th.join()
            if not errors:
                e2e_two = 2 * per_lane * n_h / (time.perf_counter() - t0)
            ctx_b.close()
        extras["pcie_inclusive"] = {
            "extract_images_per_s_from_host_memory": round(1.0 / t_host, 1),
            "extract_images_per_s_from_hbm_same_call": round(1.0 / t_dev, 1),
            "images_per_s_end_to_end": round(e2e, 1),
            "images_per_s_end_to_end_two_surveys_in_flight": None if e2e_two is None else round(e2e_two, 1),
            "end_to_end_step_seconds": {k: round(float(v), 4) for k, v in ts.items()},
            "pcie_gbytes_per_s_of_pixels": round(e2e * h * w * 3 / 1e9, 1),
            "note": f"{n_h} views in page-locked host memory ({h * w * 3 / 1e6:.0f} MB of BGR each), uploaded in chunks of 25 images by the launch "
                    "sequence that extracts them, so one sequence's upload runs under the others' kernels; images_per_s_end_to_end "
                    f"is MEASURED: {n_e2e} steps (load with the uploads, link, relax; the relax of a step under the next step's load as in "
                    f"the timed region, the last one waited for) over {n_h} cameras of the survey"}
        release()
    except Exception as ex:
        extras["pcie_inclusive"] = {"error": str(ex)}
    try:
        gg, _, _ = pipeline.run(ctx, grid, images, shape, start_ori)
        # (b) the same 50-camera group the CPU leg relaxes, on the device
        sub = np.arange(min(50, grid.n_images))
        e50 = gg.edges_flat(sub)
        pk50 = host.pack_edges(e50)
        t0 = time.perf_counter()
        r50 = host.relax_ground_plane(ctx, grid.position[sub], start_ori[sub], grid.model, sub, start_ori[sub], pk50)
        t50 = time.perf_counter() - t0
        extras["relax_device_50_cameras"] = {"cameras": int(len(sub)), "residual_blocks": int(r50["residual_blocks"]),
                                             "lm_iterations": int(r50["iterations_total"]), "seconds": round(t50, 4),
                                             "device_seconds": round(r50["device_s"], 4),
                                             "lm_iters_per_s": round(r50["iterations_total"] / max(r50["device_s"], 1e-9), 1)}
        # (c) what the pipeline states after INITIAL_PROCESSING run on this survey (pipeline.cpp:666-707): RelaxStage with
        # floor(n / 50) spectral groups, {ORIENTATION, GROUND_MESH} on the minimal mesh seeded from the plane
        plane = gg.relax(ctx, start_ori, host.relax_options("ORIENTATION", "GROUND_PLANE"))
        seed_mesh = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
        t0 = time.perf_counter()
        ms = gg.relax_stage(ctx, host.relax_options("ORIENTATION", "GROUND_MESH"), 0.1, previous=seed_mesh)
        tms = time.perf_counter() - t0
        errm = pipeline.orientation_errors(gg.orientations(), grid.orientation)
        extras["relax_stage_ground_mesh"] = {
            "groups": int(ms["groups"]), "seconds": round(tms, 4), "host_setup_seconds_summed": round(ms["setup_host_s"], 4),
            "device_seconds_summed": round(ms["device_s"], 4), "lm_iterations": int(ms["iterations_total"]),
            "residual_blocks": int(ms["residual_blocks"]), "track_blocks": int(ms["track_blocks"]),
            "two_ray_blocks": int(ms["two_ray_blocks"]), "images_per_s": round(grid.n_images / tms, 1),
            "median_orientation_error_rad_vs_truth": float(np.median(errm))}
        # (d) dense guided matching (densifyMesh, dense_stereo.cpp:66-403) over the same survey: every dense feature's ray
        # against the relaxed ground, the descriptor search in a 150 px disc on the device, tracks -> 3-D points
        ground = host.rebuild_mesh(grid.position, minimal=True)
        ga = ground.arrays()
        gv = ga["vertices"].copy()
        gv[:, 2] = grid.plane[0] * gv[:, 0] + grid.plane[1] * gv[:, 1]
        ground.set(gv, ga["edges"])
        # (a first call fills the context's page-locked staging pool - 0.9 GB for this survey's index -, like the warm-up steps of
        # the timed region; the second is the one reported, `first_call_seconds` beside it)
        warm = host.Surface().set(gv, ga["edges"])
        t0 = time.perf_counter()
        gg.densify_mesh(ctx, warm)
        tds_first = time.perf_counter() - t0
        ctx.profile_reset()
        t0 = time.perf_counter()
        ds = gg.densify_mesh(ctx, ground)
        tds = time.perf_counter() - t0
        _, kd_ms = ctx.profile_get(5)   # OCHIP_K_DENSE
        cloud = ground.clouds()[-1] if ds["points"] else np.zeros((0, 3))
        dz = cloud[:, 2] - (grid.plane[0] * cloud[:, 0] + grid.plane[1] * cloud[:, 1])
        extras["dense_guided_matching"] = {
            "images": ds["images"], "dense_features": ds["dense_features"], "queries": ds["queries"], "matches": ds["matches"],
            "tracks": ds["tracks"], "points": ds["points"], "seconds": round(tds, 4), "first_call_seconds": round(tds_first, 4),
            "seconds_by_phase": {"index": round(ds["index_s"], 4), "rays_and_mesh_walk_host": round(ds["rays_s"], 4),
                                 "nearest_cameras_predictions_search_unions_device_incl_pcie": round(ds["device_s"], 4),
                                 "tracks_host": round(ds["tracks_s"], 4)},
            "search_accept_union_kernel_ms": round(kd_ms, 3),
            "queries_per_s_kernel": round(ds["queries"] / max(kd_ms * 1e-3, 1e-9), 1),
            "median_abs_height_error_m": float(np.median(np.abs(dz))) if len(dz) else None}
        gg.close()
    except Exception as ex:
        extras["relax_extras_error"] = str(ex)
    try:
        # (e) the reference's real schedule of INITIAL_PROCESSING beside the headline's one batch: the survey in batches of 100
        # that start without orientations; step k extracts batch k, links batch k - 1 against everything loaded before it and
        # relaxes batch k - 2 as ONE group with two rings of context cameras, the three stages' runners side by side
        # (pipeline.cpp:522-570; host.InitialProcessing); the new cameras are bootstrapped one solve each (relax.cpp:52-80) by
        # one resident launch on the device (csrc/relax_chain.hip).  Best of two runs (the first one warms the pools).
        best = None
        for _ in range(2):
            gi, inc = pipeline.run_initial_processing(ctx, grid, images, shape, batch=100)
            erri = pipeline.orientation_errors(gi.orientations(), grid.orientation)
            gi.close()
            if best is None or inc["seconds"] < best[0]["seconds"]:
                best = (inc, erri)
        inc, erri = best
        extras["incremental_batches"] = {
            "batches": inc["batches"], "steps": inc["steps"], "images": int(n), "seconds": round(inc["seconds"], 4),
            "images_per_s": round(n / inc["seconds"], 1), "step_seconds": inc["step_seconds"],
            "load_runner_seconds": round(inc["load_runner_s"], 4), "link_runner_seconds": round(inc["link_runner_s"], 4),
            "relax_runner_seconds": round(inc["relax_runner_s"], 4), "relax_device_seconds": round(inc["relax_device_s"], 4),
            "relax_host_setup_seconds": round(inc["relax_setup_host_s"], 4),
            "lm_solves": inc["solves"], "lm_iterations": inc["lm_iterations"],
            "lm_iters_per_s": round(inc["lm_iterations"] / max(inc["relax_runner_s"], 1e-9), 1), "edges": int(inc["edges"]),
            "median_orientation_error_rad_vs_truth": float(np.median(erri)),
            "cameras_left_unoriented": int(np.sum(~np.isfinite(erri))),
            "note": "every camera starts with a NaN orientation; ONE relax group per batch (RelaxStage::init with "
                    "disable_parallelism, pipeline.cpp:545-546: round 5's figure split a batch into two groups of 50, which the "
                    "reference does not) whose cameras are bootstrapped together with the group while the graph is smaller than "
                    "twice the group (the first batches: ~2 500 LM iterations on up to 450 unknowns per batch) and one at a time "
                    "after that (relax.cpp:61-75); the three stages of consecutive batches run side by side as in the reference"}
    except Exception as ex:
        extras["incremental_batches"] = {"error": str(ex)}
    return extras


# ---------------------------------------------------------------------------------------------------------------------
# strong scaling: ONE survey over the ranks
class StrongRunner:
    """One survey per step over all ranks (parallel.survey_sharded); the relax of a survey either on rank k mod N in the
    shadow of the following surveys (pipelined) or sharded over all ranks inside the step."""

    def __init__(self, proc, args, cfg):
        from opencalibration_amd import capi, host, parallel, pipeline, synth

        self.proc, self.args = proc, args
        self.host, self.parallel, self.pipeline, self.capi = host, parallel, pipeline, capi
        self.grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)   # the same survey on every rank
        self.ctx = capi.Context(proc.device_index)
        self.lo, self.cnt = host.shard_block(self.grid.n_images, proc.rank, proc.world)
        self.images, self.shape = pipeline.synthetic_views(self.ctx, self.grid, seed=7, block=(self.lo, self.cnt))
        self.start_ori = pipeline.perturbed_orientations(self.grid, 0.1, 99)
        self.rctx = self.ctx.sibling(12)              # (created here, before any runner thread asks for a sibling)
        self.rctx.set_priority(True)                  # the latency-bound solve goes ahead of the throughput kernels
        self.pipelined = args.relax == "pipelined"
        self.k = 0                                    # surveys started (the same count on every rank)
        self.pending = None                           # this rank's relax in flight
        self.last = {}
        self.exchange = None
        self.comm = None
        if not self.pipelined:
            if proc.backend == "nccl":
                self.comm = parallel.relax_exchange_rccl(self.rctx)      # RCCL all-gathers on the solver's own stream
                self.exchange = self.comm
            else:
                self.exchange = parallel.relax_exchange()                # torch.distributed (gloo: host staging)

    def _collect(self, acc):
        if self.pending is None:
            return
        th, res, t = self.pending
        self.pending = None
        th.join()
        if getattr(th, "error", None):
            raise th.error
        self.last.update(res=res, t=t)
        if acc is not None:
            acc["relax"] = acc.get("relax", 0.0) + t["relax"]
            acc["relax_device"] = acc.get("relax_device", 0.0) + res["relax"]["device_s"]
            acc["relax_setup_host"] = acc.get("relax_setup_host", 0.0) + res["relax"]["setup_host_s"]
            acc["relax_lm_iterations"] = acc.get("relax_lm_iterations", 0.0) + res["relax"]["iterations_total"]
            acc["relaxes"] = acc.get("relaxes", 0) + 1

    def _general_engine_sharded(self):
        """(untimed, once, in a warm-up step) The flavour the reference's later pipeline states run - {ORIENTATION, GROUND_MESH},
        pipeline.cpp:582-704 - as ONE group with its residual blocks over the ranks: the general engine (relax_general.hip)
        behind the same exchange as the plane engine's.  The sharded survey leaves remote images with their 40 px subsets
        only, so the mesh flavour (which reads the feature lists) runs on a small graph every rank builds identically from
        synthetic features, and must reproduce the unsharded solve to the bit."""
        proc = self.proc
        try:
            from opencalibration_amd import synth

            grid = synth.make_grid(3, 5, feats=512, seed=17)
            g = self.host.Graph.from_synthetic(grid)
            g.link(self.ctx)
            start = self.pipeline.perturbed_orientations(grid, 0.05, 3)
            g.set_orientations(start)
            plane = g.relax(self.rctx, start, self.host.relax_options("ORIENTATION", "GROUND_PLANE"))
            seed = self.host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
            opts = self.host.relax_options("ORIENTATION", "GROUND_MESH")
            g.set_orientations(plane["orientation"])
            ref = g.relax(self.rctx, plane["orientation"], opts, 0.1, previous=seed)
            g.set_orientations(plane["orientation"])
            t2 = time.perf_counter()
            got = g.relax(self.rctx, plane["orientation"], opts, 0.1, previous=seed, shard=(proc.rank, proc.world, self.exchange))
            self.rctx.synchronize()
            dt = time.perf_counter() - t2
            g.close()
            same = bool(np.array_equal(ref["orientation"], got["orientation"]) and ref["final_cost"] == got["final_cost"] and
                        ref["iterations_total"] == got["iterations_total"])
            errm = self.pipeline.orientation_errors(got["orientation"], grid.orientation)
            return {"flavour": "ORIENTATION + GROUND_MESH, one group of 15 cameras built identically on every rank, residual blocks "
                               "over the ranks", "ranks": proc.world, "seconds": round(dt, 4),
                    "lm_iterations": int(got["iterations_total"]), "residual_blocks": int(got["residual_blocks"]),
                    "equals_the_unsharded_solve_bit_for_bit": same,
                    "median_orientation_error_rad_vs_truth": float(np.median(errm))}
        except Exception as ex:
            return {"error": repr(ex)}

    def _survey(self, owner):
        """load + link of one survey over the ranks: (graph, report, seconds, CPU seconds)"""
        _, h, w = self.shape
        c0 = time.process_time()
        t0 = time.perf_counter()
        g = self.host.Graph()
        mid = g.add_model(self.grid.model)
        st = self.parallel.survey_sharded(self.ctx, g, mid, self.grid.position, self.start_ori, self.images, w, h,
                                          edges_to=owner)
        return g, st, time.perf_counter() - t0, time.process_time() - c0

    def step(self, acc, survey=None):
        proc = self.proc
        owner = self.k % proc.world if self.pipelined else None
        self.k += 1
        g, st, t_link, cpu_link = survey.result() if survey is not None else self._survey(owner)
        res = dict(edges=g.num_edges, survey=st)
        t = {}
        if self.pipelined:
            if owner == proc.rank:
                self._collect(acc)                    # one relax in flight per rank

                def work(g=g, res=res, t=t):
                    try:
                        self.pipeline.relax_step(self.rctx, g, self.start_ori, res, t)
                        g.close()                     # (the feature records: freed off the main thread)
                    except Exception as ex:           # surfaces in _collect()
                        threading.current_thread().error = ex

                th = threading.Thread(target=work)
                th.start()
                self.pending = (th, res, t)
            else:
                g.close()
        else:
            t1 = time.perf_counter()
            rel = g.relax_ground_plane(self.rctx, self.start_ori, shard=(proc.rank, proc.world, self.exchange))
            self.rctx.synchronize()
            t["relax"] = time.perf_counter() - t1
            res["relax"] = rel
            if acc is None and "general_engine_sharded" not in self.last:
                self.last["general_engine_sharded"] = self._general_engine_sharded()
            g.close()
            self.pending = (threading.Thread(target=lambda: None), res, t)
            self.pending[0].start()
            self._collect(acc)
        if acc is not None:
            sec = st["seconds"]
            for k, v in (("load_link", t_link), ("extract", sec["extract"]), ("block_linked", sec["block_linked"]),
                         ("exchange", st["exchange_s"]), ("subsets_import", sec["subsets_import"]),
                         ("remote_links", sec["remote_links"]), ("edges_import", sec["edges_import"]),
                         ("finalize", sec["finalize"]), ("host_cpu_load_link", cpu_link),
                         ("bytes_gathered", st.get("bytes_gathered", 0)), ("exchanges", st.get("exchanges", 0))):
                acc[k] = acc.get(k, 0.0) + v
            acc["steps"] = acc.get("steps", 0) + 1
        self.last["survey"] = st
        self.last["edges"] = max(self.last.get("edges", 0), res["edges"])

    def run(self, n_steps, acc):
        # One rank: two surveys in flight as in the weak mode (the next survey extracts beside this one's link tail).  With
        # several ranks the surveys stay one after the other: two surveys' collectives issued from two threads would have
        # to reach the communicator in the same order on every rank.
        if self.proc.world == 1 and self.pipelined and os.environ.get("OCHIP_PIPELINE_SURVEYS", "2" if self.proc.cores >= 2 else "1") != "1":
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(2) as pool:
                futures = [pool.submit(self._survey, 0) for _ in range(min(2, n_steps))]
                for step in range(n_steps):
                    f = futures.pop(0)
                    f.result()
                    if step + 2 < n_steps:
                        futures.append(pool.submit(self._survey, 0))
                    self.step(acc, survey=f)
        else:
            for _ in range(n_steps):
                self.step(acc)
        self._collect(acc)

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.ctx.synth_views_free(self.images)


def strong_report(runner, proc, args, cfg, hot_max, acc):
    """The strong-scaling numbers of one timed run, gathered over the ranks (rank 0 gets the full picture)."""
    grid, steps = runner.grid, args.steps
    per_rank = proc.gather_objects({k: (round(v / steps, 5) if isinstance(v, float) else v) for k, v in acc.items()})
    st = runner.last.get("survey", {})
    relaxes = [p for p in per_rank if p.get("relaxes")]
    lm_iters = sum(p["relax_lm_iterations"] * steps for p in relaxes)
    relax_dev = sum(p["relax_device"] * steps for p in relaxes)
    return {
        "images_per_s": round(grid.n_images * steps / hot_max, 3), "ms_per_step": round(hot_max / steps * 1e3, 3),
        "ranks": proc.world, "steps": steps,
        "relax": ("pipelined over surveys: rank k mod N relaxes survey k alone, in the shadow of the following surveys' load + link; "
                  "every relax completes inside the timed region") if runner.pipelined else
                 "sharded inside the step: residual blocks over all ranks, one all-gather of the per-pair records per evaluation; "
                 "the Cholesky factorisation of the reduced system (0.6 of an iteration's ~1.3 ms at 3 003 unknowns) is "
                 "replicated on every rank, which bounds the sharded solve's speed-up at ~1.4 x by design",
        "lm_iters_per_s_in_pipeline": round(lm_iters / max(relax_dev, 1e-9), 2),
        "block_images_rank0": runner.cnt, "pairs_in_block_rank0": st.get("pairs_in_block"),
        "pairs_across_blocks_rank0": st.get("pairs_across_blocks"), "halo_images_rank0": st.get("halo_images"),
        "seconds_per_step_per_rank": per_rank,
        "exchanges_per_step": 0 if proc.world == 1 else 2,
        "bytes_gathered_per_step": per_rank[0].get("bytes_gathered", 0),
        **({"general_engine_sharded": runner.last["general_engine_sharded"]} if "general_engine_sharded" in runner.last else {}),
    }


def strong_main(args, proc, cfg):
    os.environ.setdefault("OCHIP_BLOB_SPACING", "16")
    runner = StrongRunner(proc, args, cfg)
    capi, pipeline, ctx, grid = runner.capi, runner.pipeline, runner.ctx, runner.grid
    runner.run(args.warmup, None)
    ctx.profile_reset()
    proc.barrier()
    acc = {}
    t_begin = time.perf_counter()
    runner.run(args.steps, acc)
    proc.barrier()
    hot_max = proc.max_over_ranks(time.perf_counter() - t_begin)
    report = strong_report(runner, proc, args, cfg, hot_max, acc)
    n, h, w = runner.shape
    st = runner.last["survey"]
    roofline = device_rooflines(ctx, capi, args.steps, runner.cnt, acc["extract"] / args.steps, runner.shape, True,
                                None, runner.last.get("edges", 0))
    roofline["kernel"] += f" (rank 0's block of {runner.cnt} images per step)"
    rel = runner.last.get("res", {}).get("relax")
    cpu = None
    if proc.rank == 0 and proc.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(ctx, grid, runner.images, runner.shape, proc.threads, runner.start_ori)
    if proc.rank == 0:
        relax_info = {"lm_iters_per_s_in_pipeline": report["lm_iters_per_s_in_pipeline"],
                      "unknowns": int(3 * grid.n_images + 3)}
        if rel is not None:
            err = pipeline.orientation_errors(rel["orientation"], grid.orientation)
            relax_info.update(residual_blocks=int(rel["residual_blocks"]), lm_iterations_per_relax=int(rel["iterations_total"]),
                              median_orientation_error_rad_vs_truth=float(np.median(err)),
                              cameras_left_unconstrained=int(np.sum(err > 0.02)))
        out = {
            "metric": METRIC, "value": report["images_per_s"], "unit": "images/s", "n_gpus": proc.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": report["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "images_per_s_strong_one_survey_over_all_gpus": report["images_per_s"],
            "rccl_ranks": proc.world if proc.backend == "nccl" else 0, "collective_preflight_s": proc.preflight_s,
            "config": {"workload": f"{args.config} as ONE survey over {proc.world} rank(s) (BASELINE config C4): {grid.n_images}-image "
                                   f"synthetic aerial grid {cfg['rows']}x{cfg['cols']}, {w}x{h} rendered views resident in HBM "
                                   f"(each rank holds its block), {st['features'] / max(runner.cnt, 1):.0f} AKAZE features/image, "
                                   f"{runner.last.get('edges', 0)} edges",
                       "sharding": "extract: contiguous image blocks; link: a directed pair runs on the rank that owns the later of "
                                   "its two images (both directions together); exchanges: all-gather of the 40 px subsets, "
                                   "all-gather of the pairs' results; relax: " + report["relax"],
                       "backend": proc.backend, "host_threads_per_rank": int(os.environ["OMP_NUM_THREADS"]),
                       "usable_host_cpus": proc.cores},
            "strong_scaling": report, "relax": relax_info, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    runner.close()
    proc.finish()


# ---------------------------------------------------------------------------------------------------------------------
def weak_main(args, proc, cfg):
    rank, world = proc.rank, proc.world
    from opencalibration_amd import capi, pipeline, synth

    # blob lattice of the rendered views: 16 px apart in the 1600 px working image.  The greedy 40 px subsample of
    # match_features.cpp:8-52 saturates at about 3.5 k features on a 4000 x 3000 image (the nominal
    # "4k features entering the matcher" cannot be reached at this image size with the reference's 40 px spacing); this
    # density sits at that saturation (21 px gave 3.2 k).
    os.environ.setdefault("OCHIP_BLOB_SPACING", "16")
    grid = synth.make_grid(seed=12345 + rank, rows=cfg["rows"], cols=cfg["cols"], feats=64)  # poses + camera model
    ctx = capi.Context(proc.device_index)
    images, shape = pipeline.synthetic_views(ctx, grid, seed=7 + rank)    # resident in HBM before timing starts
    start_ori = pipeline.perturbed_orientations(grid, 0.1, 99 + rank)

    last = {}
    # load and link overlapped (the reference's pipeline overlaps the stages of consecutive batches); 0 runs them
    # one after the other, which is what the per-stage numbers of DESIGN.md were taken with
    overlap = os.environ.get("OCHIP_PIPELINE_OVERLAP", "1") != "0"

    # relax of survey k overlapped with load + link of survey k + 1: Pipeline::Impl::initial_processing runs the load, link
    # and relax runners of consecutive batches together (pipeline.cpp:543-560); here successive steps are successive surveys.
    # Every relax finishes inside the timed region (the last one is joined before the closing barrier).
    relax_overlap = overlap
    # two surveys in flight need host threads for two surveys' host phases at once: worth it from 2 CPUs per rank on since
    # the relax set-up moved to the device (2 CPUs: 2 580 against 2 450 images/s; 3 CPUs: 2 920 against 2 590; before that
    # move one in flight was faster at 2 CPUs, 1 746 against 1 660).  OCHIP_PIPELINE_SURVEYS overrides.
    surveys_in_flight = os.environ.get("OCHIP_PIPELINE_SURVEYS", "2" if proc.cores // max(world, 1) >= 2 else "1") != "1"
    rctx = ctx.sibling(12) if relax_overlap else ctx      # (created here, before any runner thread asks for a sibling)
    if relax_overlap:
        rctx.set_priority(True)                            # the latency-bound solve goes ahead of the throughput kernels

    def run_steps(n_steps, acc):
        pending = []

        def collect(entry):
            th, g, res, t = entry
            th.join()
            if getattr(th, "error", None):
                raise th.error
            last.update(res=res, t=t, link_work=last.get("link_work"))
            if not relax_overlap:
                g.close()
            if acc is not None:
                for k, v in list(t.items()) + [("link_" + k, v) for k, v in res["link_timers"].items()] + \
                        [("relax_setup_host", res["relax"]["setup_host_s"]), ("relax_device", res["relax"]["device_s"]),
                         ("relax_lm_iterations", res["relax"]["iterations_total"])]:
                    acc[k] = acc.get(k, 0.0) + v

        verbose = "bench" in os.environ.get("OCHIP_VERBOSE", "").split(",")
        # load + link of survey k + 1 beside the link tail of survey k (two host threads; the native side lets one survey
        # extract at a time and gives alternating surveys their own link contexts, csrc/host/load_link.cpp): the
        # reference runs the load runners of a batch beside the link runners of the batch before it (pipeline.cpp:543-560)
        from concurrent.futures import ThreadPoolExecutor

        in_flight = 2 if (overlap and relax_overlap and surveys_in_flight) else 1
        pool = ThreadPoolExecutor(in_flight)
        futures = [pool.submit(pipeline.run, ctx, grid, images, shape, start_ori, overlap=overlap, relax=not relax_overlap)
                   for _ in range(min(in_flight, n_steps))]
        for step in range(n_steps):
            ta = time.perf_counter()
            g, res, t = futures.pop(0).result()
            if step + in_flight < n_steps:
                futures.append(pool.submit(pipeline.run, ctx, grid, images, shape, start_ori, overlap=overlap,
                                           relax=not relax_overlap))
            tb = time.perf_counter()
            last["link_work"] = dict(g.match_work(), images=grid.n_images)
            if relax_overlap:
                if pending:
                    collect(pending.pop())           # one relax in flight at a time
                if verbose:
                    print(f"[bench] load+link {tb - ta:.3f} s, collect previous {time.perf_counter() - tb:.3f} s", file=sys.stderr)

                def work(g=g, res=res, t=t):
                    try:
                        pipeline.relax_step(rctx, g, start_ori, res, t)
                        g.close()                    # (1.8 GB of feature records: freed off the main thread)
                    except Exception as ex:          # surfaces in collect()
                        threading.current_thread().error = ex

                th = threading.Thread(target=work)
                th.start()
                pending.append((th, g, res, t))
            else:
                th = threading.Thread(target=lambda: None)
                th.start()
                collect((th, g, res, t))
        while pending:
            collect(pending.pop())
        pool.shutdown()

    run_steps(args.warmup, None)
    ctx.profile_reset()
    proc.barrier()
    acc = {}
    cpu_begin = time.process_time()
    t_begin = time.perf_counter()
    run_steps(args.steps, acc)
    proc.barrier()
    hot_max = proc.max_over_ranks(time.perf_counter() - t_begin)
    # CPU seconds of this process (all its threads) over the whole timed region: the per-stage figures of
    # stage_seconds_per_step are process-wide deltas over intervals that overlap each other in the pipelined step (relax of
    # survey k under load + link of survey k + 1) and so count each other's threads; this one counts everything once
    host_cpu_total = (time.process_time() - cpu_begin) / args.steps
    value = grid.n_images * world * args.steps / hot_max

    res = last["res"]
    rel = res["relax"]
    n, h, w = shape
    # With the load and link stages overlapped the extraction shares the device with the link kernels during the timed
    # steps, so the in-situ figure understates the extract kernels; one extra step with the stages run one after the
    # other (same process, after the timed region, HIP events as above) gives the sequence on its own.
    roofline = device_rooflines(ctx, capi, args.steps, grid.n_images, acc["extract"] / args.steps, shape, overlap,
                                last["link_work"], res["edges"])
    staged_relax = None
    if overlap and rank == 0:
        ctx.profile_reset()
        g2, res2, t2 = pipeline.run(ctx, grid, images, shape, start_ori, overlap=False)
        staged_relax = dict(res2["relax"])
        g2.close()
        n2, ms2 = ctx.profile_get(capi.K_AKAZE)
        ach2 = extract_algorithmic_bytes(w, h) * grid.n_images / t2["extract"] / 1e9
        roofline["staged"] = {"what": "one untimed step with the stages one after the other (extraction alone on the device)",
                              "achieved": round(ach2, 1), "frac": round(ach2 / 8000.0, 4),
                              "frac_with_the_pass_structure_of_rounds_2_to_4": round(extract_algorithmic_bytes(w, h, True) * grid.n_images / t2["extract"] / 8e12, 4),
                              "valu_issue": valu_issue(grid.n_images, t2["extract"]),
                              "avg_launch_ms": round(t2["extract"] * 1e3 / max(n2, 1), 3),
                              "hip_event_ms_per_sequence_overlapped": round(ms2 / max(n2, 1), 3),
                              "stage_seconds": {k: round(float(v), 4) for k, v in t2.items()}}
    err = pipeline.orientation_errors(rel["orientation"], grid.orientation)
    lm_iters = acc.get("relax_lm_iterations", 0.0)
    # LM iterations per second of the relax stage's device phase.  In the timed steps the relax of one survey runs in the
    # shadow of the next survey's extraction (shared GPU, latency-bound solve): that rate is the headline LM figure; the
    # same relax alone on the device (the staged step after the timed region) is quoted beside it
    in_pipeline = round(lm_iters / max(acc.get("relax_device", 1e-9), 1e-9), 2)
    alone = round(staged_relax["iterations_total"] / max(staged_relax["device_s"], 1e-9), 2) if staged_relax else None
    relax_info = {"lm_iters_per_s_in_pipeline": in_pipeline,
                  "lm_iters_per_s": in_pipeline,
                  "lm_iters_per_s_relax_alone_on_the_device": alone,
                  "lm_iters_per_s_note": "relax stage of the timed steps (device phase; shares the GPU with the next survey's "
                                         "extraction when the stages are pipelined); *_alone: the staged step",
                  "lm_iterations_per_step": lm_iters / args.steps,
                  "unknowns": int(3 * grid.n_images + 3), "residual_blocks": int(rel["residual_blocks"]),
                  "median_orientation_error_rad_vs_truth": float(np.median(err)),
                  "cameras_left_unconstrained": int(np.sum(err > 0.02))}

    extras = {}
    if rank == 0 and world == 1 and os.environ.get("OCHIP_BENCH_EXTRAS", "1") != "0":
        extras = beside_the_headline(ctx, grid, images, shape, start_ori, hot_max / args.steps, rctx if relax_overlap else None)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU leg is timed at N = 1 only
        cpu = cpu_baseline_leg(ctx, grid, images, shape, proc.threads, start_ori)

    # ---- N > 1: a short run of the strong mode beside the weak headline (ONE survey of this shape over the N ranks), so
    #      that a scaling run of the default command carries both curves.  A watchdog bounds it: the headline is printed
    #      even if the strong run cannot finish.
    strong = None
    ctx.synth_views_free(images)
    images = None
    if world > 1 and os.environ.get("OCHIP_BENCH_STRONG_BESIDE", "1") != "0":
        strong = run_strong_beside(args, proc, cfg)

    if rank == 0:
        out = {
            "metric": METRIC,
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(hot_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE,
            "data": "synthetic",
            # both curves by name: `value` is the weak one (a survey per GPU, the unit the reference parallelises over:
            # pipeline.cpp:42-49, 543-560); the strong one is BASELINE config C4 (ONE 1 000-image survey over the N GPUs)
            # the boundary of the reference hands over host images (cv::Mat, load_stage.cpp:35-50): the same step measured
            # from page-locked host memory, PCIe inclusive (best of one / two surveys in flight); never `value`
            "images_per_s_from_host_memory": (lambda p: None if not isinstance(p, dict) or "error" in p else
                                              max(p.get("images_per_s_end_to_end") or 0.0,
                                                  p.get("images_per_s_end_to_end_two_surveys_in_flight") or 0.0))(
                                                      (extras or {}).get("pcie_inclusive")),
            # the same survey through the reference's own schedule of INITIAL_PROCESSING (batches of 100 without orientations,
            # extract k | link k - 1 | relax k - 2 side by side, every new camera bootstrapped by a solve of its own): never `value`
            "images_per_s_reference_schedule": (lambda p: None if not isinstance(p, dict) or "error" in p else p.get("images_per_s"))(
                (extras or {}).get("incremental_batches")),
            "value_assumes": "views resident in HBM when the timed region starts (bench contract); PCIe-inclusive rate beside it",
            "images_per_s_weak_one_survey_per_gpu": round(value, 3),
            "images_per_s_strong_one_survey_over_all_gpus": (round(value, 3) if world == 1 else
                                                             (strong or {}).get("images_per_s")),
            "rccl_ranks": world if proc.backend == "nccl" else 0, "collective_preflight_s": proc.preflight_s,
            "config": {"workload": f"{args.config}: {grid.n_images}-image synthetic aerial grid {cfg['rows']}x{cfg['cols']}, "
                                   f"{w}x{h} rendered views resident in HBM, {res['features_per_image']:.0f} AKAZE "
                                   f"features/image ({res['sparse_per_image']:.0f} after the 8 px NMS), {res['edges']} edges",
                       "stages_overlapped": ("load and link (ranges of links start as soon as their images are extracted)"
                                             + ("; relax of survey k with load + link of survey k + 1, as the reference runs the "
                                                "load / link / relax runners of consecutive batches together (pipeline.cpp:543-560); "
                                                "every relax completes inside the timed region" if relax_overlap else "")
                                             + ("; the extraction of survey k + 1 starts when survey k's last image is extracted, "
                                                "beside survey k's remaining link ranges (two surveys in flight, one extracting at a "
                                                "time; every survey is linked and relaxed inside the timed region)"
                                                if relax_overlap and surveys_in_flight else ""))
                                            if overlap else "none (OCHIP_PIPELINE_OVERLAP=0)",
                       "stages_timed": ["extract: grey + INTER_AREA + AKAZE + std::sort by response / 8 px NMS / feature records "
                                        "(device), one block copy per image (host)",
                                        "link: kNN, 40px subsample (host), upload, Hamming 2-NN, ratio test + std::sort, PROSAC "
                                        "order, homography RANSAC (device), decompose + inlier lists (host)",
                                        "relax: ground-plane assembly (host) + LM with block-envelope Cholesky, all cameras in "
                                        "one group (device)"],
                       "host_threads_per_rank": int(os.environ["OMP_NUM_THREADS"]), "usable_host_cpus": proc.cores,
                       "per_rank": "one grid of this shape per GPU, no data-path collective"},
            "host_cpu_s_per_step_total": round(host_cpu_total, 4),
            "stage_seconds_per_step": {k: round(v / args.steps, 5) for k, v in acc.items()},
            "relax": relax_info,
            "beside_the_headline": extras,
            "strong_scaling": strong,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    proc.finish()


def run_strong_beside(args, proc, cfg):
    """A short strong-scaling run after the weak timed region: every rank starts `bench.py --scaling strong` as a CHILD process
    (same RANK / WORLD_SIZE, its own rendezvous port) and waits for it.  The exchanges of that mode have only ever run between
    ranks sharing one GPU before the scaling run; in a child, nothing it does - a hang (bounded by
    300 s), an abort inside a collective - can take the weak headline with it.  Returns rank 0's report
    (None on the other ranks) or {"error": ...}."""
    import subprocess

    steps = max(2, min(args.steps, 5))
    port = int(os.environ.get("MASTER_PORT", "29500")) + 17            # (the same on every rank)
    env = dict(os.environ, MASTER_PORT=str(port), OCHIP_BENCH_STRONG_BESIDE="0")
    env.pop("OCHIP_BENCH_CPUS", None)                                   # (already pinned: inherited)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(proc.world), "--scaling", "strong", "--relax", args.relax,
           "--steps", str(steps), "--warmup", "1", "--config", args.config, "--no-cpu-baseline"]
    try:
        done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=300.0)
    except subprocess.TimeoutExpired:
        return {"error": "the strong-scaling run did not finish in time"}
    except OSError as ex:
        return {"error": repr(ex)}
    if proc.rank != 0:
        return None
    if done.returncode != 0:
        return {"error": f"the strong-scaling run exited with code {done.returncode}"}
    for line in reversed(done.stdout.decode(errors="replace").splitlines()):
        if line.startswith("{"):
            try:
                child = json.loads(line)
            except ValueError:
                break
            report = child.get("strong_scaling") or {}
            report.update(images_per_s=child.get("value"), ms_per_step=child.get("ms_per_step"), steps=child.get("steps"),
                          how="a child process per rank after the weak timed region: bench.py --scaling strong")
            return report
    return {"error": "the strong-scaling run printed no line"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", help="C1|C2|C3 (BASELINE.md §3); C3 = the 1 000-image grid of the metric")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one survey per GPU (default); strong = ONE survey over the N GPUs (config C4)")
    ap.add_argument("--relax", default="pipelined", choices=["pipelined", "sharded"],
                    help="strong scaling only: pipelined = the relax of survey k runs on rank k mod N in the shadow of the next "
                         "surveys (as at N = 1); sharded = inside the step, residual blocks over all ranks, RCCL exchange")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # (no launcher around us: be the launcher)
    proc = Proc(args)
    from opencalibration_amd import synth

    cfg = synth.CONFIGS[args.config]
    if args.scaling == "strong":
        strong_main(args, proc, cfg)
    else:
        weak_main(args, proc, cfg)


if __name__ == "__main__":
    main()
