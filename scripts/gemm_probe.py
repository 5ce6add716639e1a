import torch, time
dev = torch.device("cuda")
for (B, m, n) in ((16384, 552, 501), (4096, 552, 501), (16384, 501, 501)):
    X = torch.randn(B, m, dtype=torch.float64, device=dev); A = torch.randn(m, n, dtype=torch.float64, device=dev)
    for _ in range(3): Y = X @ A
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): Y = X @ A
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(B, m, n, "%.3f ms  %.1f TFLOP/s" % (dt * 1e3, 2 * B * m * n / dt / 1e12))
