import time, torch
dev = torch.device("cuda:0")
x = torch.zeros((256 << 20) // 4, device=dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(3):
        for _ in range(20): x.add_(1.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): x.add_(1.0)          # ~20 x 100 us
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); g.replay(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("first launch %.1f us, second launch (first still running) %.1f us, total %.1f us" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t0) * 1e6))
print("x[0] =", float(x[0]))
