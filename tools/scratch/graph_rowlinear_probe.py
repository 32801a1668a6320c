import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from memotr_amd.modules.linear import _RowLinear, row_linear, long_linear, configure_blas
from memotr_amd.functions import clip_ops
configure_blas()
dev = torch.device("cuda")
torch.manual_seed(0)
def probe(name, fn, rows, K, N):
    w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    def f(x, w, b):
        return fn(x, w, b)
    x = torch.randn(rows, K, device=dev, requires_grad=True)
    ws, bs = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    g = torch.cuda.make_graphed_callables(f, (x, ws, bs))
    out = []
    for it in range(4):
        xi = torch.randn(rows, K, device=dev, requires_grad=True)
        wi = (w + 0.01 * it).requires_grad_(True); bi = (b + it).requires_grad_(True)
        go = torch.randn(rows, N, device=dev)
        y = g(xi, wi, bi); y.backward(go)
        xr = xi.detach().clone().requires_grad_(True); wr = wi.detach().clone().requires_grad_(True); br = bi.detach().clone().requires_grad_(True)
        yr = fn(xr, wr, br); yr.backward(go)
        rel = lambda a, c: float((a - c).norm() / (c.norm() + 1e-9))
        out.append("y %.1e gx %.1e gw %.1e gb %.1e" % (rel(y, yr), rel(xi.grad, xr.grad), rel(wi.grad, wr.grad), rel(bi.grad, br.grad)))
    print(name, rows, K, N); [print("   ", o) for o in out]
for rows in (300, 3060):
    probe("rowlinear", lambda x, w, b: _RowLinear.apply(x, w, b, False), rows, 256, 256)
    probe("rowlinear+relu", lambda x, w, b: _RowLinear.apply(x, w, b, True), rows, 256, 256)
    probe("F.linear", lambda x, w, b: F.linear(x, w, b), rows, 256, 256)
    probe("rowlinear 256->768", lambda x, w, b: _RowLinear.apply(x, w, b, False), rows, 256, 768)
