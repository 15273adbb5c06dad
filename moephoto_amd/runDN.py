"""DN plugin table and option builder -- mirror of python/runDN.py:9-38 (SEDN and NetDN rows; the
MPRNet / NAFNet / VSR_Cleaning rows are other model families, out of this engine's scope)."""
import os

from .config import config
from .imageProcess import initModel, Option
from .models import NetDN, SEDN
from .runSR import engineRamCoef

# key -> (weights path, constructor, squeeze dim, padding, align)
mode_switch = {
    '15': ('./model/l15/model_new.pth', SEDN, 1, 7, 8),
    '25': ('./model/l25/model_new.pth', SEDN, 1, 7, 8),
    '50': ('./model/l50/model_new.pth', SEDN, 1, 7, 8),
    'lite5': ('./model/dn_lite5/model_new.pth', NetDN, 1, 7, 8),
    'lite10': ('./model/dn_lite10/model_new.pth', NetDN, 1, 7, 8),
    'lite15': ('./model/dn_lite15/model_new.pth', NetDN, 1, 7, 8),
}


def getOpt(optDN):
    model = optDN['model']
    path, ctor, sd, padding, align = mode_switch[model]
    opt = Option(os.path.join(config.modelRoot, path))
    opt.modelDef, opt.padding, opt.align = ctor, padding, align
    opt.strength = optDN.get('strength', 1.0)
    opt.cropsize = config.getConfig()[1 if model[:4] == 'lite' else 2]
    opt.modelCached = initModel(opt, opt.model, 'DN' + model)
    opt.ramCoef = engineRamCoef(opt.modelCached, 1)
    if sd:
        opt.fixChannel = 0
        opt.squeeze = lambda x: x.squeeze(sd)
        opt.unsqueeze = lambda x: x.unsqueeze(sd)
    return opt
