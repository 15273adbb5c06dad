"""Import the MoePhoto reference (read-only at /root/reference) in THIS container only.

Test-infrastructure helper used by tools/gen_golden.py to generate golden vectors.
Nothing here ships: /root/reference does not exist on the GPU box.

Shims:
  1. fake `torchvision.transforms.functional.to_tensor` + empty `torchvision.ops`
     (imageProcess.py:12, models.py:9-12)
  2. fake `gevent` (progress.py:4)
  3. `torch.load(..., weights_only=False)` for the legacy-format zoo (imageProcess.py:306)
The tile grid is pinned by overriding `config.calcFreeMem` (config.py:61) and the crop sizes.
"""
import os
import sys
import types

REF = os.environ.get('MOE_REFERENCE', '/root/reference')


def _install_shims():
    import numpy as np
    import torch

    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvf = types.ModuleType('torchvision.transforms.functional')
    tvo = types.ModuleType('torchvision.ops')

    def to_tensor(pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t.to(torch.float32)

    tvf.to_tensor = to_tensor
    tvt.functional = tvf
    tv.transforms = tvt
    tv.ops = tvo
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.transforms', tvt)
    sys.modules.setdefault('torchvision.transforms.functional', tvf)
    sys.modules.setdefault('torchvision.ops', tvo)

    gv = types.ModuleType('gevent')

    class _G:
        def __init__(self, f, *a):
            self.f, self.a = f, a

        def start(self):
            self.f(*self.a)

    gv.spawn = lambda f, *a: _G(f, *a)
    gv.sleep = lambda *a: None
    gv.idle = lambda *a: None
    sys.modules.setdefault('gevent', gv)

    _load = torch.load

    def load(f, *a, **k):
        k['weights_only'] = False
        return _load(f, *a, **k)

    torch.load = load


_ref = None


def load_reference(crop_sr=0, crop_dn=0, crop_dns=0, free_mem=1 << 40):
    """Returns a namespace with the reference modules; chdirs into the reference root
    (mode_switch paths are relative: runSR.py:11)."""
    global _ref
    if _ref is None:
        _install_shims()
        os.chdir(REF)
        sys.path.insert(0, os.path.join(REF, 'python'))
        from config import config
        import imageProcess
        import models
        import MoeNet_lite2
        import runSR
        import runDN
        _ref = types.SimpleNamespace(config=config, imageProcess=imageProcess, models=models,
                                     MoeNet_lite2=MoeNet_lite2, runSR=runSR, runDN=runDN)
    c = _ref.config
    c.crop_sr, c.crop_dn, c.crop_dns = crop_sr, crop_dn, crop_dns
    c.calcFreeMem = lambda *a, **k: int(free_mem)
    return _ref
