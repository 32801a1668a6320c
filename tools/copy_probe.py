import time, torch
x = torch.zeros(320, 256, device="cuda"); y = torch.zeros_like(x)
big = torch.zeros(22323, 256, device="cuda"); bigy = torch.zeros_like(big)
xt = torch.zeros(256, 320, device="cuda").t()
def t(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    return h / n * 1e6, w / n * 1e6
print("copy_ contiguous 320x256 (hipMemcpyAsync D2D): host %.1f us wall %.1f us" % t(lambda: y.copy_(x)))
print("copy_ contiguous 22323x256:                     host %.1f us wall %.1f us" % t(lambda: bigy.copy_(big)))
print("copy_ from a transposed view (kernel):          host %.1f us wall %.1f us" % t(lambda: y.copy_(xt)))
print("add_ (kernel):                                  host %.1f us wall %.1f us" % t(lambda: y.add_(1.0)))
print("torch.add(x, 0, out=y) as a copy (kernel):      host %.1f us wall %.1f us" % t(lambda: torch.add(x, 0.0, out=y)))
print("zeros(320,256):                                 host %.1f us wall %.1f us" % t(lambda: torch.zeros(320, 256, device='cuda')))
print("empty(320,256):                                 host %.1f us wall %.1f us" % t(lambda: torch.empty(320, 256, device='cuda')))
idx = torch.arange(10, device="cuda")
print("index_select 10 rows:                           host %.1f us wall %.1f us" % t(lambda: x.index_select(0, idx)))
print("cat of two:                                     host %.1f us wall %.1f us" % t(lambda: torch.cat((x, y), 0)))
