import time, torch, numpy as np
torch.cuda.set_device(0)
torch.zeros(1).cuda()
def t(tag, fn, n=10):
    fn(0)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(n): fn(i+1)
    torch.cuda.synchronize()
    print("%-60s %.2f ms" % (tag, 1e3*(time.perf_counter()-t0)/n), flush=True)
rng=np.random.default_rng(0)
sizes=[int(v) for v in rng.integers(9000,12500,16)]
srcs=[torch.randn(s,257) for s in sizes]
t("pin_memory() of ~11 MB, varying sizes", lambda i: srcs[i%16].pin_memory())
t("pin_memory() + to(cuda, non_blocking)", lambda i: srcs[i%16].pin_memory().to("cuda", non_blocking=True))
keep=[]
t("same, keeping the pinned tensors alive", lambda i: keep.append(srcs[i%16].pin_memory().to("cuda", non_blocking=True)))
t("to(cuda) from pageable", lambda i: srcs[i%16].to("cuda"))
buf=torch.empty(13000*257).pin_memory()
def via_buf(i):
    s=srcs[i%16]; v=buf[:s.numel()].view_as(s); v.copy_(s); return v.to("cuda", non_blocking=True)
t("copy into ONE reused pinned buffer + to(cuda, non_blocking)", via_buf)
t("torch.tensor(list, device=cuda) x5", lambda i: [torch.tensor(list(range(32)), dtype=torch.int64, device="cuda") for _ in range(5)])
t("torch.zeros(500,32,257,cuda)", lambda i: torch.zeros(500,32,257,device="cuda"))
import torch.multiprocessing as mp
sh=[s.clone().share_memory_() for s in srcs]
t("pin_memory() of shared-memory tensors", lambda i: sh[i%16].pin_memory())
