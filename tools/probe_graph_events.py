"""Can a kernel INSIDE a replayed hipGraph be bracketed by HIP events?  torch refuses external events on ROCm, HIP itself
has hipEventRecordWithFlags(hipEventRecordExternal): an event-record NODE in the captured graph."""
import ctypes, torch
dev = torch.device("cuda:0")
torch.zeros((1,), device=dev)
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]      # the runtime torch itself loaded
print("runtime:", path)
hip = ctypes.CDLL(path)
a = torch.randn((4096, 4096), device=dev); b = torch.randn((4096, 4096), device=dev)
s = torch.cuda.Stream()
def ev():
    e = ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(e)) == 0
    return e
e0, e1 = ev(), ev()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    c = a @ b
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        c = a @ b
        r0 = hip.hipEventRecordWithFlags(e0, ctypes.c_void_p(s.cuda_stream), 1)
        c = a @ b
        r1 = hip.hipEventRecordWithFlags(e1, ctypes.c_void_p(s.cuda_stream), 1)
        c = a @ b
print("record rc", r0, r1)
for i in range(5):
    g.replay(); torch.cuda.synchronize()
    ms = ctypes.c_float()
    rc = hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
    print("replay %d: rc %d bracketed matmul %.4f ms" % (i, rc, ms.value))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); c = a @ b; t1.record(); torch.cuda.synchronize(); print("eager matmul %.4f ms" % t0.elapsed_time(t1))
