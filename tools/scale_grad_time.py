import sys, os, torch, ctypes
sys.path.insert(0, '/root/repo')
import ctc_amd
from ctc_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
n = 150*256*158
g = torch.randn(n, device=dev)
for val in (1.0, 0.5):
    go = torch.tensor(val, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    gr = torch.cuda.CUDAGraph()
    for _ in range(3): lib.ctc_amd_scale_grad(g.data_ptr(), go.data_ptr(), n, s)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(50): lib.ctc_amd_scale_grad(g.data_ptr(), go.data_ptr(), n, st.cuda_stream)
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("grad_out = %.1f: %.2f us per scale_grad launch (24 MB gradient)" % (val, e0.elapsed_time(e1) * 1e3 / 500))
