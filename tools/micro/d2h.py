import torch, time
x = torch.empty(64*1024*1024//4, device='cuda')
for mb in (8, 64):
    n = mb*1024*1024//4
    h = torch.empty(n, pin_memory=True)
    for _ in range(3): h.copy_(x[:n], non_blocking=True); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(10): h.copy_(x[:n], non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(f"D2H pinned {mb} MB: {dt*1e3:.3f} ms = {mb/1024/dt:.1f} GB/s")
    hp = torch.empty(n)
    t=time.perf_counter()
    for _ in range(5): hp.copy_(x[:n]); 
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print(f"D2H pageable {mb} MB: {dt*1e3:.3f} ms = {mb/1024/dt:.1f} GB/s")
