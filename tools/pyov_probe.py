import time, torch, sys
sys.path.insert(0, "/root/repo")
import fps_amd
from fps_amd import _lib
x = torch.zeros(1000, dtype=torch.float64, device="cuda")
g = torch.zeros(1000, dtype=torch.float64, device="cuda")
N = 20000
t0 = time.perf_counter()
for _ in range(N): _lib.producer_stream(x, g, None, None, None)
t1 = time.perf_counter()
for _ in range(N): int(torch.cuda.current_stream(x.device).cuda_stream)
t2 = time.perf_counter()
for _ in range(N): (_lib.ptr(x), _lib.ptr(g), _lib.ptr(None), _lib.ptr(None), _lib.ptr(None))
t3 = time.perf_counter()
print(f"producer_stream {1e6*(t1-t0)/N:.2f} us, Stream object {1e6*(t2-t1)/N:.2f} us, 5 x ptr {1e6*(t3-t2)/N:.2f} us")
assert _lib.producer_stream(x) == int(torch.cuda.current_stream(x.device).cuda_stream)
