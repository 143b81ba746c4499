import torch, time
x = torch.zeros(64, device="cuda")
y = torch.zeros(32 * 1024 * 1024, device="cuda", dtype=torch.float16)  # 64 MB
for name, fn in (("tiny add_", lambda: x.add_(1)), ("64MB fill", lambda: y.fill_(1.0))):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1), "us per launch")
# graph replay of 1000 tiny kernels
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(1000): x.add_(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("graph tiny add_", e0.elapsed_time(e1), "us per launch")
