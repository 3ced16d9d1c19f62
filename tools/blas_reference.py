"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, f32) reaches on the shapes of the
training products, for comparison with gemm_kernel.h.  Prints TFLOP/s per shape."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
R = 393216


def bench(a, b, iters=10):
    for _ in range(3):
        a @ b
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        a @ b
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


shapes = {"fwd1 [R,64]x[64,512]": (R, 64, 512), "fwd2 [R,512]x[512,256]": (R, 512, 256), "delta1 [R,256]x[256,512]": (R, 256, 512)}
for name, (m, k, n) in shapes.items():
    a = torch.randn(m, k, device=dev)
    b = torch.randn(k, n, device=dev)
    t = bench(a, b)
    print(f"{name}: {t * 1e6:.0f} us, {2 * m * k * n / t / 1e12:.1f} TFLOP/s")
# weight gradients: [512,R]x[R,256] and [64,R]x[R,512]
for name, (m, k, n) in {"dw2 [512,R]x[R,256]": (512, R, 256), "dw1 [64,R]x[R,512]": (64, R, 512)}.items():
    a = torch.randn(k, m, device=dev).t()
    b = torch.randn(k, n, device=dev)
    t = bench(a, b)
    print(f"{name}: {t * 1e6:.0f} us, {2 * m * k * n / t / 1e12:.1f} TFLOP/s")
