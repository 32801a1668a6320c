import time, torch, torch.nn.functional as F
import torch.cuda.tunable as tn
def bench(fn, n=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(n): fn()
    host=(time.perf_counter()-t0)/n*1e6
    torch.cuda.synchronize()
    tot=(time.perf_counter()-t0)/n*1e6
    return host, tot
x=torch.randn(1,310,256,device="cuda"); w=torch.randn(256,256,device="cuda"); b=torch.randn(256,device="cuda")
x2=torch.randn(310,256,device="cuda")
print("default      F.linear host/total us", bench(lambda: F.linear(x,w,b)))
print("default      mm       host/total us", bench(lambda: x2@w))
torch.backends.cuda.preferred_blas_library("cublas")
print("rocblas pref mm       host/total us", bench(lambda: x2@w))
print("rocblas pref F.linear host/total us", bench(lambda: F.linear(x,w,b)))
tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(5); tn.set_max_tuning_iterations(20); tn.set_filename("/tmp/tun.csv", False)
F.linear(x,w,b); x2@w; torch.cuda.synchronize()
print(tn.get_results())
tn.tuning_enable(False)
print("tuned        F.linear host/total us", bench(lambda: F.linear(x,w,b)))
print("tuned        mm       host/total us", bench(lambda: x2@w))
y=torch.randn(1,317,256,device="cuda")
print("tunable-on, untuned shape F.linear host/total us", bench(lambda: F.linear(y,w,b)))
