import os, sys
ROOT = '/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops
dev = torch.device('cuda:0'); _lib.init(0)
g = torch.Generator().manual_seed(0)
for M, N, K in [(2100, 1540, 40), (2100, 1540, 72), (2100, 1540, 200), (2100, 1540, 232), (2100, 1540, 48), (2100, 1540, 56), (2100, 1540, 64), (2048, 1536, 200), (2048, 1536, 40)]:
    A = torch.randn(M, K, generator=g).to(dev); B = torch.randn(N, K, generator=g).to(dev)
    out = torch.zeros(M, N, device=dev)
    ops.gemm(A, B, False, True, None, 0, out=out, beta=0)
    ref = A.double() @ B.double().t()
    err = (out.double() - ref).abs()
    print(M, N, K, 'rel %.3g' % float(err.max() / ref.abs().max()), 'argmax', divmod(int(err.argmax()), N))
