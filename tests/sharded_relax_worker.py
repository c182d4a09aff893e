"""Worker of tests/test_gpu_relax_sharded.py: launched by torch.distributed.run with 2 ranks (gloo) that share
cuda:0.  Every rank links the same synthetic graph, relaxes it unsharded, then sharded (residual blocks split over
the ranks, per-pair records exchanged through parallel.relax_exchange) and checks the two agree to the bit.
OCHIP_TEST_BACKEND=nccl with ONE rank runs the same through the RCCL branch of the exchange (all_gather_into_tensor on the
device buffers of the solver): all the 1-GPU boxes allow, but it is the real transport on the real buffers.
OCHIP_TEST_EXCHANGE=rccl takes libochip's own communicator instead (ochip_rccl_*: ncclAllGather on the solver's stream)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, parallel, synth  # noqa: E402


def main():
    dist.init_process_group(os.environ.get("OCHIP_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = capi.Context(0)
    grid = synth.make_grid(3, 5, feats=512, seed=17)
    g = host.Graph.from_synthetic(grid)
    g.link(ctx)
    rng = np.random.default_rng(3)
    axes = rng.normal(size=(grid.n_images, 3))
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    start = synth.quat_mul(grid.orientation, np.concatenate([axes * np.sin(0.05), np.full((grid.n_images, 1), np.cos(0.05))], 1))
    g.set_orientations(start)
    ref = g.relax_ground_plane(ctx, start)
    g.set_orientations(start)
    native = os.environ.get("OCHIP_TEST_EXCHANGE") == "rccl"
    exchange = parallel.relax_exchange_rccl(ctx) if native else parallel.relax_exchange()
    got = g.relax_ground_plane(ctx, start, shard=(rank, world, exchange))
    if native:
        stats = exchange.stats()
        exchange.close()
        assert stats["exchanges"] >= got["iterations_total"] and stats["bytes_gathered"] > 0, stats
        print("RCCL_EXCHANGES", stats, flush=True)
    same = np.array_equal(ref["orientation"], got["orientation"]) and np.array_equal(ref["plane"], got["plane"])
    same = same and ref["iterations_total"] == got["iterations_total"] and ref["final_cost"] == got["final_cost"]
    err = 2 * np.arccos(np.clip(np.abs(np.sum(got["orientation"] * grid.orientation, axis=1)), 0, 1))
    flags = [None] * world
    dist.all_gather_object(flags, (bool(same), float(err.max()), int(got["residual_blocks"])))
    if rank == 0:
        print("SHARDED_RELAX", "OK" if all(f[0] for f in flags) and flags[0][1] < 5e-3 else "MISMATCH", flags, flush=True)
    g.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
