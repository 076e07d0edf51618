"""Kernel micro-benchmarks through the C ABI (GPU box): GEMM shapes of the hot path and the fused
projection + cross-entropy forward/backward.  Prints achieved fp32 TFLOP/s per launch (random data).
usage: python scripts/bench_kernels.py [gemm] [ce]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops
dev = torch.device('cuda:0'); _lib.init(0)


def timeit(f, reps=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def bench_gemm():
    shapes = [  # (label, M, N, K, transA, transB)
        ('square 4096 NT', 4096, 4096, 4096, 0, 1),
        ('square 4096 NN', 4096, 4096, 4096, 0, 0),
        ('square 4096 TN', 4096, 4096, 4096, 1, 0),
        ('S  chunk  NT  1280x20000x200', 1280, 20000, 200, 0, 1),
        ('S  full   NT  6400x20000x200', 6400, 20000, 200, 0, 1),
        ('dX chunk  NN  1280x200x20000', 1280, 200, 20000, 0, 0),
        ('dX full   NN  6400x200x20000', 6400, 200, 20000, 0, 0),
        ('dW chunk  TN  20000x200x1280', 20000, 200, 1280, 1, 0),
        ('dW full   TN  20000x200x6400', 20000, 200, 6400, 1, 0),
        ('xproj     NT  6400x1024x200', 6400, 1024, 200, 0, 1),
        ('dWih      TN  1024x200x6400', 1024, 200, 6400, 1, 0),
        ('dxin      NN  6400x200x1024', 6400, 200, 1024, 0, 0),
        ('dec dWih  TN  2048x400x6400', 2048, 400, 6400, 1, 0),
        ('dec dWih  TN  2048x400x3000', 2048, 400, 3000, 1, 0),
        ('dec dWhh  TN  2048x512x6400', 2048, 512, 6400, 1, 0),
        ('dec dX    NN  6400x400x2048', 6400, 400, 2048, 0, 0),
        ('dec dX    NN  3000x400x2048', 3000, 400, 2048, 0, 0),
        ('dec xproj NT  6400x2048x400', 6400, 2048, 400, 0, 1),
        ('eval hWhh NT 16000x1024x256', 16000, 1024, 256, 0, 1),
        ('eval hWhh NT 16000x768x256', 16000, 768, 256, 0, 1),
        ('eval Wc   NT 16000x200x256', 16000, 200, 256, 0, 1),
    ]
    for lab, M, N, K, ta, tb in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        t = timeit(lambda: ops.gemm(A, B, bool(ta), bool(tb), out=C))
        print('gemm %-34s %8.1f us  %6.1f TF' % (lab, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)


def bench_ce():
    V, E = 20000, 200
    for N, frac in ((6400, 1.0), (6400, 0.47), (12800, 0.47), (1 << 16, 0.47)):
        X = torch.randn(N, E, device=dev) * 0.3
        W = torch.randn(V, E, device=dev) * 0.1
        b = torch.randn(V, device=dev) * 0.1
        T, R = N // 32, 32
        tgt = torch.randint(1, V, (T, R), device=dev)
        tgt[torch.rand(T, R, device=dev) > frac] = 0
        eos = tgt.t().contiguous()
        Xr = X.view(T, R, E).clone().requires_grad_(True)
        Wp, bp = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        live = int((tgt != 0).sum())

        def fwd():
            with torch.no_grad():
                return ops.proj_ce(Xr, Wp, bp, eos, 0)
        t = timeit(fwd)
        print('ce fwd N=%6d live=%6d %8.1f us  %6.1f TF (all rows) %6.1f TF (live rows)' % (
            N, live, t * 1e6, 2.0 * N * V * E / t / 1e12, 2.0 * live * V * E / t / 1e12), flush=True)

        def fb():
            Xr.grad = Wp.grad = bp.grad = None
            ops.proj_ce(Xr, Wp, bp, eos, 0).sum().backward()
        t2 = timeit(fb, reps=5)
        print('ce fwd+bwd                    %8.1f us  %6.1f TF (all rows) %6.1f TF (live rows)' % (
            t2 * 1e6, 8.0 * N * V * E / t2 / 1e12, 8.0 * live * V * E / t2 / 1e12), flush=True)


if __name__ == '__main__':
    what = sys.argv[1:] or ['gemm', 'ce']
    if 'gemm' in what:
        bench_gemm()
    if 'ce' in what:
        bench_ce()
