"""SR plugin table and option builder -- mirror of python/runSR.py:9-48 with the engine-backed
constructors in the table's constructor slot.  `getOpt({'model': 'a', 'scale': 4, 'ensemble': 0})`
returns an Option whose `modelCached` runs on the HIP engine; `sr(opt)` wraps the self-ensemble."""
import ctypes
import os

from . import _lib
from .config import config
from .imageProcess import ensemble, initModel, Option
from .models import Net2x, Net3x, Net4x
from .MoeNet_lite2 import Net

# key -> (weights path relative to config.modelRoot, constructor); memory coefficients are derived from the
# engine itself (engineRamCoef) instead of the reference's measured table (runSR.py:9)
mode_switch = {
    'a2': ('./model/a2/model_new.pth', Net2x),
    'a3': ('./model/a3/model_new.pth', Net3x),
    'a4': ('./model/a4/model_new.pth', Net4x),
    'p2': ('./model/p2/model_new.pth', Net2x),
    'p3': ('./model/p3/model_new.pth', Net3x),
    'p4': ('./model/p4/model_new.pth', Net4x),
    'lite2': ('./model/lite/model.pth', Net),
    'lite4': ('./model/lite/model_4.pth', lambda: Net(upscale=4)),
    'lite8': ('./model/lite/model_8.pth', lambda: Net(upscale=8)),
}


def engineRamCoef(model, scale, ref=128):
    """0.9 / (device bytes per input pixel-plane): workspace of one ref x ref plane plus its fp32 result tile.
    Plays the role of the reference's ramCoef rows (bytes per pixel measured with test/memTest.py)."""
    ws = _lib.check(_lib.lib().moe_net_workspace_bytes(model._h, 1, ref, ref))
    per_px = ws / float(ref * ref) + 4.0 * scale * scale
    return 0.9 / per_px


sr = lambda opt: (lambda x: ensemble(opt)(x) / (opt.ensemble + 1)) if opt.ensemble else ensemble(opt)


def getOpt(optSR):
    """python/runSR.py:30-48: step dict {'model': 'a', 'scale': 4, 'ensemble': 0} -> Option, or None for an unknown model/scale pair.
    Colour planes become the batch (dim-1 squeeze / unsqueeze), padding 9 for the x3 nets and 5 otherwise, cropsize = config.crop_sr."""
    key = '{}{}'.format(optSR['model'], optSR['scale'])
    entry = mode_switch.get(key)
    if entry is None:
        return None
    rel_path, ctor = entry
    opt = Option(os.path.join(config.modelRoot, rel_path))
    opt.mode, opt.scale, opt.modelDef = optSR['model'], optSR['scale'], ctor
    opt.fixChannel = 0
    opt.squeeze, opt.unsqueeze = (lambda t: t.squeeze(1)), (lambda t: t.unsqueeze(1))
    opt.padding = {3: 9}.get(opt.scale, 5)
    ens = optSR.get('ensemble')
    opt.ensemble = ens if isinstance(ens, int) and 0 <= ens <= 7 else config.ensembleSR
    opt.cropsize = config.getConfig()[0]
    opt.modelCached = initModel(opt, opt.model, 'SR' + key)
    opt.ramCoef = engineRamCoef(opt.modelCached, opt.scale)
    return opt
