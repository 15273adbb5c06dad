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
    opt = Option()
    opt.mode = optSR['model']
    opt.scale = optSR['scale']
    nmode = opt.mode + str(opt.scale)
    if nmode not in mode_switch:
        return None
    opt.fixChannel = 0
    opt.squeeze = lambda x: x.squeeze(1)
    opt.unsqueeze = lambda x: x.unsqueeze(1)
    opt.padding = 9 if opt.scale == 3 else 5
    opt.model = os.path.join(config.modelRoot, mode_switch[nmode][0])
    opt.modelDef = mode_switch[nmode][1]
    opt.ensemble = optSR['ensemble'] if 'ensemble' in optSR and (0 <= optSR['ensemble'] <= 7) else config.ensembleSR
    opt.cropsize = config.getConfig()[0]
    opt.modelCached = initModel(opt, opt.model, 'SR' + nmode)
    opt.ramCoef = engineRamCoef(opt.modelCached, opt.scale)
    return opt
