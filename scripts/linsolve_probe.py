import torch, time
dev = torch.device("cuda", 0)
for B, N in ((1024, 440), (1024, 256), (16384, 160), (4096, 200), (16384, 1054)):
    try:
        if B * N * N * 8 > 150e9: raise RuntimeError("too big")
        A = torch.randn(B, N, N, dtype=torch.float64, device=dev) + N ** 0.5 * torch.eye(N, dtype=torch.float64, device=dev)
        b = torch.randn(B, N, 1, dtype=torch.float64, device=dev)
        x = torch.linalg.solve(A, b); torch.cuda.synchronize()
        t0 = time.perf_counter(); x = torch.linalg.solve(A, b); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"solve B={B} N={N}: {dt*1e3:.1f} ms  ({B*2/3*N**3/dt/1e12:.2f} TF/s)  resid {float((A@x-b).abs().max()):.1e}", flush=True)
    except Exception as e:
        print("B", B, "N", N, "failed:", str(e)[:80], flush=True)
