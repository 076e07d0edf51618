"""cvae_mi_static -- the entry point that allocates nothing (its [K,B] intermediates live in a static device
array of the library) -- against cvae_mi, the existing entry point whose workspace the caller brings.
Same kernels, so the bar is bit-exact.  Two calls in a row, the second smaller (the array then still holds the
first call's values past the second call's extent); then the size limit and its refusal."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(K, B, Z, seed, dev):
    g = torch.Generator().manual_seed(seed)
    mu, sg = torch.randn(B, Z, generator=g), torch.rand(B, Z, generator=g) + 0.3
    z = mu + sg * torch.randn(K, B, Z, generator=g)
    return z.to(dev), mu.to(dev), sg.to(dev)


def test_mi_static_matches_mi_on_consecutive_calls():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from cvae_amd import _lib, ops
    _lib.init(0)
    dev = torch.device('cuda:0')
    limit = _lib.query('cvae_mi_static_max_rows')
    assert limit == 131072
    for K, B, Z, seed in ((100, 513, 32, 1), (10, 33, 32, 2), (4, 2, 8, 3), (256, 512, 16, 4)):   # last: K*B == limit
        z, mu, sg = _case(K, B, Z, seed, dev)
        want = ops.mi(z, mu, sg)
        out = torch.full((B,), float('nan'), device=dev)
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats(dev)['allocation.all.allocated']
        free_before = torch.cuda.mem_get_info(dev)[0]          # hipMemGetInfo: sees a hipMalloc / hipFree inside the library
        got = ops.mi_static(z, mu, sg, out=out)
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats(dev)['allocation.all.allocated'] == before      # torch allocated nothing either
        assert torch.cuda.mem_get_info(dev)[0] == free_before                          # ... and neither did the library
        assert got.data_ptr() == out.data_ptr() and torch.equal(got, want), (K, B)
    # the call is legal under stream capture, where any allocation or synchronisation fails: one captured call,
    # replayed on new inputs written into the captured buffers
    z, mu, sg = _case(100, 513, 32, 7, dev)
    out = torch.empty((513,), device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.mi_static(z, mu, sg, out=out)                       # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.mi_static(z, mu, sg, out=out)
    z2, mu2, sg2 = _case(100, 513, 32, 8, dev)
    z.copy_(z2); mu.copy_(mu2); sg.copy_(sg2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ops.mi(z2, mu2, sg2))
    z, mu, sg = _case(257, 512, 8, 5, dev)                                             # one sample past the limit
    with pytest.raises(_lib.CvaeHipError, match='static scratch limit'):
        ops.mi_static(z, mu, sg)
