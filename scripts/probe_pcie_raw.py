import torch, time
torch.cuda.init()
for mb in (225, 900):
    n = mb << 20
    src = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
    dst = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
    for streams in (1, 2, 4):
        ss = [torch.cuda.Stream() for _ in range(streams)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 8
        for r in range(reps):
            for i, s in enumerate(ss):
                with torch.cuda.stream(s):
                    dst[i].copy_(src[i], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mb} MB chunks, {streams} stream(s): {reps * streams * n / dt / 1e9:.1f} GB/s")
