#!/usr/bin/env python3
"""Run one of the reference's own scripts, byte-for-byte unchanged, on the MI355X path.

    python scripts/run_reference.py /path/to/reference/Coupled-VAE train.py [script args ...]
    python scripts/run_reference.py /path/to/reference/Coupled-CVAE test.py

What it does (and nothing else): puts coupled-vae_amd/dropin/<tree> AHEAD of the reference tree on sys.path --
Python itself would put the script's directory first, which is why PYTHONPATH alone cannot do this -- so that
`from Modules.vae_couple import VAE_COUPLE` etc. (CV/Experiment.py:11-27) resolve to the drop-in classes while
every module the accelerated path does not replace still comes from the reference's own files; sets
COUPLED_VAE_REFERENCE for that fall-through; changes into the tree (the reference reads data/ and writes
outputs/ relative to the working directory, CV/config.py:512-516); and runs the script as __main__.
`utils.utils` stays the reference's own file: none of its functions is on the accelerated path any more."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    if len(argv) < 3:
        sys.exit(__doc__)
    tree = os.path.abspath(argv[1])
    name = os.path.basename(tree.rstrip(os.sep))
    dropin = os.path.join(ROOT, 'coupled-vae_amd', 'dropin', name)
    if not os.path.isdir(dropin):
        sys.exit('no drop-in package for %r (expected one of: %s)' % (
            name, ', '.join(sorted(d for d in os.listdir(os.path.dirname(dropin)) if not d.startswith('_')))))
    script = argv[2] if os.path.isabs(argv[2]) else os.path.join(tree, argv[2])
    if not os.path.isfile(script):
        sys.exit('no such script: %s' % script)
    os.environ['COUPLED_VAE_REFERENCE'] = os.path.dirname(tree)
    os.chdir(tree)
    sys.path[:0] = [dropin, os.path.join(ROOT, 'coupled-vae_amd'), tree]
    sys.argv = [script] + argv[3:]
    pin_per_tensor_optimizer()
    runpy.run_path(script, run_name='__main__')


def pin_per_tensor_optimizer():
    """Experiment.train lists the shared embedding once per alias (CV/Experiment.py:589-600) and Experiment.step
    hands that list to clip_grad_norm_ and Adam (:509-522).  Written against per-tensor loops, that means: the alias
    gradient is scaled and stepped once per listing, one after the other (what the reference's CPU run records and
    cvae_clip_adam reproduces).  torch >= 2 on a GPU defaults both calls to multi-tensor kernels, which apply the
    in-place update to every listing of the same tensor CONCURRENTLY -- a data race with an undefined result.  The
    launcher therefore makes foreach=False the default of the two names Experiment.py imports
    (`from torch.optim import Adam`, `from torch.nn.utils import clip_grad_norm_`); an explicit foreach= still wins.
    tests/test_gpu_reference_pattern.py drives the mirror through exactly these two calls against the fixtures."""
    import functools
    import torch

    class Adam(torch.optim.Adam):
        def __init__(self, params, *a, **k):
            k.setdefault('foreach', False)
            super().__init__(params, *a, **k)

    torch.optim.Adam = Adam
    _clip = torch.nn.utils.clip_grad_norm_

    @functools.wraps(_clip)
    def clip_grad_norm_(*a, **k):
        k.setdefault('foreach', False)
        return _clip(*a, **k)

    torch.nn.utils.clip_grad_norm_ = clip_grad_norm_


if __name__ == '__main__':
    main(sys.argv)
