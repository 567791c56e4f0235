import torch
dev=torch.device("cuda:0")
def timeit(fn,reps=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for mb in (268, 1073):
    n=mb*1000*1000//4
    a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
    t=timeit(lambda: a.zero_()); print("memset %4d MB: %.1f us  %.2f TB/s"%(mb,t,mb/t/1e0*1e-6*1e6/1e6))
    t=timeit(lambda: b.copy_(a)); print("copy   %4d MB: %.1f us  %.2f TB/s (r+w)"%(mb,t,2*mb/t))
    t=timeit(lambda: a.sum()); print("read   %4d MB: %.1f us  %.2f TB/s"%(mb,t,mb/t))
