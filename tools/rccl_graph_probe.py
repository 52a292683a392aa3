import os, torch, torch.distributed as dist, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.ones(1 << 20, device="cuda")
y = torch.zeros(1 << 20, device="cuda")
dist.all_reduce(x)            # warm up the communicator outside capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y.copy_(x); dist.all_reduce(y)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        y.copy_(x * 2.0)
        dist.all_reduce(y)
        w = dist.all_reduce(y, async_op=True)
        w.wait()
        y.mul_(0.5)
    torch.cuda.synchronize()
    x.fill_(3.0)
    g.replay(); torch.cuda.synchronize()
    print("RCCL collectives captured and replayed in a HIP graph: OK, y[0] =", y[0].item())
except Exception as e:
    print("RCCL graph capture FAILED:", type(e).__name__, e)
dist.destroy_process_group()
