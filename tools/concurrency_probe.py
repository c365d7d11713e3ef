"""Do kernels on two HIP streams run concurrently on this box? (environment probe)
A: a spin kernel of few workgroups on stream 1; B: the same on stream 2.  Concurrent -> wall ~ max, serial -> sum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x1 = torch.zeros(1 << 14, device=dev); x2 = torch.zeros(1 << 14, device=dev)

def work(x, n):
    for _ in range(n):
        x.add_(1.0)              # tiny kernels: a chain of 16 K-element adds (one workgroup wave each)

def run(par, n=2000):
    torch.cuda.synchronize(); t = time.perf_counter()
    if par:
        with torch.cuda.stream(s1): work(x1, n)
        with torch.cuda.stream(s2): work(x2, n)
    else:
        with torch.cuda.stream(s1): work(x1, n); work(x2, n)
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3

big1 = torch.randn(8192, 8192, device=dev); big2 = torch.randn(8192, 8192, device=dev)
def mm(par, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    if par:
        with torch.cuda.stream(s1):
            for _ in range(n): torch.sin_(x1.view(-1)[:256].repeat(1)) if False else big1.mul_(1.0001)
        with torch.cuda.stream(s2):
            for _ in range(n): big2.mul_(1.0001)
    else:
        with torch.cuda.stream(s1):
            for _ in range(n): big1.mul_(1.0001)
            for _ in range(n): big2.mul_(1.0001)
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
for _ in range(2): run(True, 200); run(False, 200)
print("launch-bound chains: serial %.2f ms  two streams %.2f ms" % (run(False), run(True)))
# graph-captured chains remove the host from the picture
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1): work(x1, 2000)
with torch.cuda.graph(g2): work(x2, 2000)
def graphs(par):
    torch.cuda.synchronize(); t = time.perf_counter()
    if par:
        with torch.cuda.stream(s1): g1.replay()
        with torch.cuda.stream(s2): g2.replay()
    else:
        with torch.cuda.stream(s1): g1.replay(); g2.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
graphs(True); graphs(False)
print("graph chains of 2000 tiny kernels: serial %.2f ms  two streams %.2f ms" % (graphs(False), graphs(True)))
