import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
x = torch.randn(2, 256, 64, 64, dtype=torch.float64); w = torch.randn(256, 256, 3, 3, dtype=torch.float64)
x.requires_grad_(True); w.requires_grad_(True)
for n in (torch.get_num_threads(), 64, 32, 16, 8):
    torch.set_num_threads(n)
    for rep in range(2):
        t = time.perf_counter()
        y = torch.nn.functional.conv2d(x, w, padding=1); y.sum().backward()
        dt = time.perf_counter() - t
    print("threads", n, "fwd+bwd %.3f s" % dt, flush=True)
