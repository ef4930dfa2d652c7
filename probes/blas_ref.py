import torch, time
def bench(M, N, K, dt=torch.bfloat16, it=10):
    a = torch.randn(M, K, device="cuda", dtype=dt); w = torch.randn(N, K, device="cuda", dtype=dt)
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): c = a @ w.t()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / it
    print(f"M={M} N={N} K={K} {dt}: {t*1e3:.3f} ms = {2*M*N*K/t/1e12:.0f} TFLOP/s", flush=True)
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (150784, 6144, 768), (150784, 2304, 768), (150784, 768, 3072), (150784, 1536, 768),
                  (37696, 2048, 256), (37696, 1536, 256), (37696, 256, 1024), (37696, 512, 512)]:
    bench(M, N, K)
bench(150784, 6144, 768, torch.float32, 3)
