"""Pipeline builder for the SR / DN steps -- the callers of the hot path (python/procedure.py:46-73,109-136,156-201).

`genProcess(steps)` turns a MoePhoto step list such as

    [{'op': 'file'}, {'op': 'DN', 'model': 'lite5', 'strength': 1.0}, {'op': 'SR', 'model': 'a', 'scale': 2}]

into one callable: image in (HWC uint8/uint16 numpy array, a file when the first step is 'file', or a raw video frame
`(bytes, height, width)` when it is {'op': 'buffer', 'bitDepth': 16} -- python/video.py:23, procedure.py:141-142) -> image out, with
everything between the upload (toTorch) and the download (toOutput) resident on the device: DN through RGBFilter
(python/procedure.py:52-55), SR through runSR.sr (:63-73), resize through moe_resize (:104-107), output = toFloat -> toOutput (:128-136).  The progress/ETA
nodes of the reference are observability only and are not reproduced; `nodes` lists the resolved
steps.  Ops other than file / buffer / DN / SR / resize / output belong to other model families and raise.
"""
from functools import reduce

from . import runDN, runSR
from .config import config
from .imageProcess import RGBFilter, apply, readFile, resize, toBuffer, toFloat, toNumPy, toOutput, toTorch, writeFile

stepOpts = dict(SR={'toInt': ['scale', 'ensemble'], 'getOpt': runSR}, DN={'toFloat': ['strength'], 'getOpt': runDN},
                resize={'toInt': ['width', 'height'], 'toFloat': ['scaleW', 'scaleH']})


def convertValues(T, o, keys):
    for key in keys:
        if key in o:
            o[key] = T(o[key])


class Context(object):
    imageMode = 'RGB'
    palette = None


def genProcess(steps, bitDepth=8, outFile=None):
    steps = [dict(s) for s in steps]
    ctx = Context()
    funcs, nodes = [], []
    has_file = bool(steps) and steps[0]['op'] == 'file'
    has_buffer = bool(steps) and steps[0]['op'] == 'buffer'
    if has_file:
        funcs.append(readFile(context=ctx))
    if has_buffer:     # video frames: raw bgr24 / bgr48le in, the same out (channel order is irrelevant: planes are independent)
        bitDepth = int(steps[0].get('bitDepth', 16))
        funcs.append(toNumPy(bitDepth))
    funcs.append(toTorch(bitDepth, config.dtype(), config.device()))
    for opt in steps:
        op = opt['op']
        if op in ('file', 'buffer', 'output'):
            continue
        if op not in stepOpts:
            raise NotImplementedError('op "{}" is not part of the SR/DN hot path this engine implements'.format(op))
        so = stepOpts[op]
        convertValues(int, opt, so.get('toInt', []))
        convertValues(float, opt, so.get('toFloat', []))
        if op == 'resize':      # procResize (python/procedure.py:104-107)
            funcs.append(resize(opt, dict(source=has_buffer)))
            nodes.append(dict(op='resize', mode=opt['method']))
            continue
        o = so['getOpt'].getOpt(opt)
        if o is None:
            raise ValueError('unknown model for step {}'.format(opt))
        opt['opt'] = o
        if op == 'SR':
            if not opt['scale'] > 1:
                raise TypeError('Invalid scale setting for SR.')
            funcs.append(runSR.sr(o))
        else:
            funcs.append(RGBFilter(o))
        nodes.append(dict(op=op, model=opt.get('model'), scale=opt.get('scale', 1)))
    funcs += [toFloat, toOutput(bitDepth)]
    if has_file and outFile is not None:
        funcs.append(lambda im: writeFile(im, outFile, ctx))
    if has_buffer:
        funcs.append(toBuffer(bitDepth))
        run = lambda im: reduce(apply, funcs, im)
        return (lambda frame: [] if not frame[0] else [run(frame)]), nodes     # a list of buffers per frame (procedure.py:122-125)
    return (lambda im: reduce(apply, funcs, im)), nodes


def runFrames(process, read, write, width, height, bitDepth=16, start=0, stop=-1):
    """The per-frame loop of SR_vid (python/video.py:349-360) without the ffmpeg plumbing: `read(nbytes)` yields raw frames of
    width*height*3 samples, every frame from `start` on goes through `process` (a genProcess 'buffer' pipeline) and each
    returned buffer is handed to `write`.  Returns the number of frames written."""
    frameBytes = width * height * 3 * (1 if bitDepth <= 8 else 2)
    i = n = 0
    while stop < 0 or i <= stop:
        raw = read(frameBytes)
        if len(raw) == 0:
            break
        if len(raw) != frameBytes:
            raise ValueError('short frame: {} of {} bytes'.format(len(raw), frameBytes))
        if i >= start:
            for buf in process((raw, height, width)):
                if buf:
                    write(buf)
                    n += 1
        i += 1
    return n
