import torch
dev = torch.device("cuda:0")
n = 134217728 // 2   # 134 MB of bf16
ring = 6
xs = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(ring)]
ys = [torch.empty_like(x) for x in xs]
for k in range(2):
    for i in range(ring): ys[i].copy_(xs[i])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(6 * ring): ys[i % ring].copy_(xs[i % ring])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (6 * ring)
print(f"copy 134 MB -> 134 MB: {us:.1f} us, {2*134.2/us*1e3:.0f} GB/s total")
e0.record()
for i in range(6 * ring): ys[i % ring].zero_()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (6 * ring)
print(f"fill 134 MB: {us:.1f} us, {134.2/us*1e3:.0f} GB/s")
e0.record()
for i in range(6 * ring): s = xs[i % ring].view(torch.int16).max()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (6 * ring)
print(f"read-reduce 134 MB: {us:.1f} us, {134.2/us*1e3:.0f} GB/s")
