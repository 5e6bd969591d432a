import torch, time
dev=torch.device("cuda",0)
for mb in (16, 6.5, 64):
    n=int(mb*1e6)
    h=torch.empty(n,dtype=torch.uint8).pin_memory(); d=torch.empty(n,dtype=torch.uint8,device=dev)
    for name,fn in (("H2D",lambda: d.copy_(h,non_blocking=True)),("D2H",lambda: h.copy_(d,non_blocking=True))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts=[]
        for _ in range(20):
            t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
        t=sorted(ts)[len(ts)//2]
        print(f"{name} {mb} MB pinned: {t*1e3:.3f} ms = {n/t/1e9:.1f} GB/s")
