"""Oracle: the uint8/uint16 <-> float edges of the pipeline.  TEST INFRASTRUCTURE.

  to_float_image  toTorch    python/imageProcess.py:259-263  (HWC uint8 -> CHW /255; >8 bit: /2^bits)
  to_output       toOutput   python/imageProcess.py:245-257  (x * 2^bits, clamp [0, 2^bits-1], TRUNCATE)
  to_hwc          toFloat    python/imageProcess.py:238-243
"""
import numpy as np


def to_float_image(img, bit_depth=8):
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    chw = np.ascontiguousarray(a.transpose(2, 0, 1))
    if bit_depth <= 8:
        # torchvision to_tensor: uint8 -> float32 / 255 (a true division, not a multiply by 1/255)
        return chw.astype(np.float32) / np.float32(255)
    return chw.astype(np.float32) / np.float32(1 << bit_depth)


def to_hwc(x):
    return np.ascontiguousarray(np.asarray(x, np.float32).transpose(1, 2, 0))


def to_output(img_hwc, bit_depth=8):
    quant = 1 << bit_depth
    v = np.asarray(img_hwc, np.float32) * np.float32(quant)
    v = np.clip(v, 0, quant - 1)
    dtype = np.uint8 if bit_depth <= 8 else (np.int16 if bit_depth <= 15 else np.int32)
    return v.astype(dtype)  # C-style truncation toward zero, like torch .to(uint8)
