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
    """python/runDN.py:25-38: step dict {'model': 'lite5', 'strength': 1.0} -> Option.  cropsize = config.crop_dn for the lite nets,
    crop_dns for SEDN; planes become the batch along the table's squeeze dim."""
    name = optDN['model']
    rel_path, ctor, sq_dim, padding, align = mode_switch[name]
    opt = Option(os.path.join(config.modelRoot, rel_path))
    opt.modelDef, opt.padding, opt.align = ctor, padding, align
    opt.strength = optDN.get('strength', 1.0)
    crops = config.getConfig()
    opt.cropsize = crops[1] if name.startswith('lite') else crops[2]
    opt.modelCached = initModel(opt, opt.model, 'DN' + name)
    opt.ramCoef = engineRamCoef(opt.modelCached, 1)
    if sq_dim:
        opt.fixChannel = 0
        opt.squeeze, opt.unsqueeze = (lambda t: t.squeeze(sq_dim)), (lambda t: t.unsqueeze(sq_dim))
    return opt
