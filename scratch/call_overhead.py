import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine, _lib
dev = torch.device('cuda')
x = torch.randn(1024, device=dev); y = torch.rand(1024, device=dev)
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    el = time.perf_counter() - t0; torch.cuda.synchronize()
    return el / n * 1e6
lib = _lib.load(); s = torch.cuda.current_stream().cuda_stream; px, py = x.data_ptr(), y.data_ptr()
print("raw ctypes yr_sigmoid      : %.1f us/call (host)" % t(lambda: lib.yr_sigmoid(px, 1024, s)))
print("engine.sigmoid_            : %.1f us/call (host)" % t(lambda: engine.sigmoid_(x)))
print("engine.sigmoid_bwd (alloc) : %.1f us/call (host)" % t(lambda: engine.sigmoid_bwd(x, y)))
print("torch.sigmoid_             : %.1f us/call (host)" % t(lambda: x.sigmoid_()))
print("torch.cuda.current_stream(): %.1f us/call" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("_cuda_getCurrentRawStream(0): %.2f us/call" % t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("current_device()+raw        : %.2f us/call" % t(lambda: torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())))
assert torch._C._cuda_getCurrentRawStream(0) == torch.cuda.current_stream().cuda_stream
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    assert torch._C._cuda_getCurrentRawStream(0) == st.cuda_stream
print("raw stream tracks torch.cuda.stream(): ok")
