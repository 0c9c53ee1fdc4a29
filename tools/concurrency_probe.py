import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1] if len(sys.argv) > 1 else "16"
import torch
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev) for _ in range(32)]
torch.cuda._sleep(1000); torch.cuda.synchronize()
# calibrate
t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize(); one = time.perf_counter() - t0
print("one sleep kernel: %.3f ms" % (one * 1e3))
for n in (1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i]):
            torch.cuda._sleep(20_000_000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("queues=%s streams=%2d: %.3f ms -> %.1f kernels at a time" % (os.environ["GPU_MAX_HW_QUEUES"], n, dt * 1e3, n * one / dt))
