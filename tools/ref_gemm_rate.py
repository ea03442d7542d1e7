"""Reference point only (nothing here is used by the product): what the vendor fp32 GEMM behind torch.matmul (rocBLAS / hipBLASLt)
reaches on the Swin linear shapes of the B = 16 step, next to this build's gemm_dma kernel (tools/bench_gemm.py)."""
import torch

dev = 'cuda'
torch.backends.cuda.matmul.allow_tf32 = False
for name, k, n in (('qkv', 256, 768), ('proj', 256, 256), ('fc1', 256, 1024), ('fc2', 1024, 256)):
    m = 16 * 72 * 72
    a = torch.randn(m, k, device=dev)
    w = torch.randn(k, n, device=dev)
    bias = torch.randn(n, device=dev)
    for _ in range(5):
        torch.addmm(bias, a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.addmm(bias, a, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'vendor fp32 addmm {name} M={m} {k}->{n}: {ms * 1e3:7.1f} us  {2.0 * m * k * n / ms / 1e9:6.1f} TF', flush=True)
