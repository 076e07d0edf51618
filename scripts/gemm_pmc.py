"""Diagnostic: a few launches of the long-K GEMM shapes of the CE backward, for rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops
dev = torch.device('cuda:0'); _lib.init(0)
shapes = [(3072, 200, 20000, 0, 0), (20000, 200, 3072, 1, 0), (6400, 2048, 400, 0, 1), (2048, 400, 6400, 1, 0),
          (6400, 400, 2048, 0, 0), (2048, 512, 6400, 1, 0)]
for M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(A, B, bool(ta), bool(tb), out=C)
    torch.cuda.synchronize()
