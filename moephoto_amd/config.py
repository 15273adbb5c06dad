"""Runtime configuration -- the small part of the reference's `config` singleton the hot path reads
(python/config.py:21-98, python/defaultConfig.py:2-23): dtype()/device()/getRunType()/getConfig()/
calcFreeMem().  Free memory comes from hipMemGetInfo (through torch.cuda.mem_get_info) instead of
NVML (python/readgpu.py)."""
import torch

from . import _lib


class Config(object):
    def __init__(self):
        self.deviceId = 0
        self.cuda = True            # this engine has no CPU path; kept for interface parity
        self.fp16 = True
        self.crop_sr = 'auto'
        self.crop_dn = 'auto'
        self.crop_dns = 'auto'
        self.ensembleSR = 0
        self.maxGraphicMemoryUsage = 0
        self.modelRoot = '.'        # directory that holds ./model/<name>/model_new.pth
        self.tilesPerBatch = 0      # 0: engine default

    def getConfig(self):
        g = lambda v: 0 if v == 'auto' else int(v)
        return g(self.crop_sr), g(self.crop_dn), g(self.crop_dns)

    def dtype(self):
        return torch.half if self.cuda and self.fp16 else torch.float

    def device(self):
        if not self.cuda:
            raise _lib.EngineError('moephoto_amd has no CPU path (config.cuda must stay True)')
        return torch.device('cuda:{}'.format(self.deviceId))

    def getRunType(self):
        return 2 if self.fp16 else 1

    def getFreeMem(self, emptyCache=False):
        if emptyCache:
            torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info(self.deviceId)
        return free - 2 ** 28

    def calcFreeMem(self, ratio=.9, emptyCache=False):
        # The engine allocates its workspace and tile pools with hipMalloc, outside torch's caching allocator: memory torch has
        # reserved but not handed out is NOT available to it (the reference adds it back because its nets allocate through torch,
        # python/config.py:61-71).  Releasing torch's cache costs a device synchronisation and the next frames' allocations, so it is
        # done only on request -- by the caller whose plan or workspace did not fit (imageProcess._plan_for retries once with it).
        free = self.getFreeMem(emptyCache=emptyCache) * ratio
        if self.maxGraphicMemoryUsage > 0:
            free = min(free, self.maxGraphicMemoryUsage * 2 ** 20 - torch.cuda.memory_allocated(self.deviceId))
        return int(free)


config = Config()
