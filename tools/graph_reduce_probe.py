"""Does torch's multi-block reduction survive hipGraph replay?  (round 3 finding: bias gradients of captured linears
came back as garbage from replay 1 on.)"""
import torch
torch.manual_seed(0)
dev = torch.device("cuda")
for rows, cols in ((3060, 256), (66969, 256), (300, 256), (66969, 2048)):
    x = torch.randn(rows, cols, device=dev)
    filler = torch.randn(1 << 20, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            y = x.sum(0); z = (filler * 2).sum()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = x.sum(0)
        t = filler * 2          # a later allocation that may reuse the reduction's scratch
        z = t.sum()
        w = (x * 1.5).sum(0)
    errs = []
    for it in range(4):
        x.copy_(torch.randn(rows, cols, device=dev)); filler.copy_(torch.randn(1 << 20, device=dev))
        g.replay(); torch.cuda.synchronize()
        ref = x.double().sum(0)
        errs.append((float((y.double() - ref).abs().max()), float((w.double() - 1.5 * ref).abs().max()),
                     float((z.double() - 2 * filler.double().sum()).abs())))
    print(rows, cols, ["%.2e/%.2e/%.2e" % e for e in errs])
