"""hipGraph memset / memcpy nodes: are they ordered with the kernels around them on replay?

    python tools/graph_memset_probe.py                                   # ROCm 7.2 default: memset wrong from replay 1 on
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_memset_probe.py  # all zeros

Each graph is  A: buf += 1 | memset(buf, 0) or memcpy(buf <- zeros) | B: out = buf + x | C: buf += 3 ; a correct
replay leaves out == x.  Printed: max |out - x| per replay."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda")
for n in (256, 1 << 16, 1 << 22):
    buf = torch.zeros(n, device=dev); x = torch.randn(n, device=dev); out = torch.empty(n, device=dev)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        buf.add_(1); out.copy_(buf + x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.add_(1.0)                                                    # kernel A: dirties the buffer
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(n * 4), ctypes.c_void_p(st))
        assert rc == 0, rc
        torch.add(buf, x, out=out)                                       # kernel B: must see zeros
        buf.add_(3.0)                                                    # kernel C: dirties it again
    res = []
    for it in range(5):
        x.copy_(torch.randn(n, device=dev)); g.replay(); torch.cuda.synchronize()
        res.append(float((out - x).abs().max()))
    print("memset node", n, res)

for n in (256, 1 << 16, 1 << 22):
    buf = torch.zeros(n, device=dev); x = torch.randn(n, device=dev); out = torch.empty(n, device=dev)
    zeros = torch.zeros(n, device=dev)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        buf.add_(1); buf.copy_(zeros); out.copy_(buf + x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.add_(1.0)
        buf.copy_(zeros)                                                 # a device-to-device memcpy node
        torch.add(buf, x, out=out)
        buf.add_(3.0)
    res = []
    for it in range(5):
        x.copy_(torch.randn(n, device=dev)); g.replay(); torch.cuda.synchronize()
        res.append(float((out - x).abs().max()))
    print("memcpy node", n, res)
