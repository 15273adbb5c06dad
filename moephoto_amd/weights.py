"""Reader/writer for the on-disk model zoo: torch's *legacy* (pre-zip) serialization.

The reference loads `model/*/model_new.pth` and `model/lite/model*.pth` with
`torch.load(path, map_location='cpu')` (python/imageProcess.py:304-307).  Those files are the
legacy layout:

    pickle(magic 0x1950a86a20f9469cfc6c) pickle(protocol 1001) pickle(sys_info)
    pickle(state_dict)          tensors = _rebuild_tensor_v2(persistent-id storage, offset, size, stride, ...)
    pickle([storage keys])
    per key:  int64 numel, then numel little-endian elements

`torch.load`'s default (weights_only=True, torch>=2.6) rejects them.  This module parses them
itself with a *closed* unpickler: the only globals it resolves are the three the zoo uses
(collections.OrderedDict, torch._utils._rebuild_tensor_v2, torch.<T>Storage), each mapped to a
local stand-in, and the four auxiliary pickles (magic, protocol, sys_info, storage keys) go through an unpickler
that refuses every global -- so no arbitrary code can run -- and torch is not needed to read weights.  Tensor views are
bounds-checked against their storage before they are materialised.
Returns `OrderedDict[str, np.ndarray]` (fp32, C-contiguous) -- what `load_state_dict` of the
engine-backed modules in `moephoto_amd.models` consumes.
"""
import io
import pickle
import struct
from collections import OrderedDict

import numpy as np

MAGIC_NUMBER = 0x1950a86a20f9469cfc6c
PROTOCOL_VERSION = 1001

_STORAGE_DTYPES = {
    'FloatStorage': np.dtype('<f4'), 'HalfStorage': np.dtype('<f2'), 'DoubleStorage': np.dtype('<f8'),
    'LongStorage': np.dtype('<i8'), 'IntStorage': np.dtype('<i4'), 'ShortStorage': np.dtype('<i2'),
    'ByteStorage': np.dtype('u1'), 'CharStorage': np.dtype('i1'), 'BoolStorage': np.dtype('?'),
}


class _StorageType:
    def __init__(self, name):
        self.name = name
        self.dtype = _STORAGE_DTYPES[name]


class _LazyStorage:
    def __init__(self, stype, key, numel):
        self.stype, self.key, self.numel = stype, key, numel
        self.data = None


class _LazyTensor:
    def __init__(self, storage, offset, size, stride):
        self.storage, self.offset, self.size, self.stride = storage, int(offset), tuple(size), tuple(stride)

    def materialize(self):
        base = self.storage.data
        if base is None:
            raise ValueError('storage {} has no data in the file'.format(self.storage.key))
        if len(self.size) == 0:
            if not 0 <= self.offset < base.size:
                raise ValueError('scalar view of storage {} is out of range'.format(self.storage.key))
            return np.array(base[self.offset], dtype=base.dtype)
        # size / stride / offset come from the file: the view must stay inside the storage (as_strided checks nothing)
        if len(self.stride) != len(self.size) or self.offset < 0 or any(int(d) < 0 for d in self.size) or any(int(t) < 0 for t in self.stride):
            raise ValueError('tensor view of storage {} has a negative size, stride or offset'.format(self.storage.key))
        if any(int(d) == 0 for d in self.size):
            return np.zeros(self.size, dtype=base.dtype)
        last = self.offset + sum((int(d) - 1) * int(t) for d, t in zip(self.size, self.stride))
        if last >= base.size:
            raise ValueError('tensor view of storage {} reaches element {} of {}'.format(self.storage.key, last, base.size))
        v = np.lib.stride_tricks.as_strided(
            base[self.offset:], shape=self.size,
            strides=tuple(int(t) * base.dtype.itemsize for t in self.stride), writeable=False)
        return np.array(v)   # writable, C-contiguous copy


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    return _LazyTensor(storage, storage_offset, size, stride)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


class _ZooUnpickler(pickle.Unpickler):
    def __init__(self, f):
        super().__init__(f)
        self.storages = {}

    def find_class(self, module, name):
        if module == 'collections' and name == 'OrderedDict':
            return OrderedDict
        if module == 'torch._utils' and name == '_rebuild_tensor_v2':
            return _rebuild_tensor_v2
        if module == 'torch._utils' and name == '_rebuild_parameter':
            return _rebuild_parameter
        if module == 'torch' and name in _STORAGE_DTYPES:
            return _StorageType(name)
        raise pickle.UnpicklingError('global {}.{} is not allowed in a model-zoo file'.format(module, name))

    def persistent_load(self, pid):
        if not isinstance(pid, tuple) or pid[0] != 'storage':
            raise pickle.UnpicklingError('unexpected persistent id {!r}'.format(pid))
        _, stype, key, _location, numel = pid[:5]
        view = pid[5] if len(pid) > 5 else None
        if view is not None:
            raise pickle.UnpicklingError('storage views are not supported')
        if key not in self.storages:
            self.storages[key] = _LazyStorage(stype, key, int(numel))
        return self.storages[key]


class _PlainUnpickler(pickle.Unpickler):
    """For the file's four auxiliary pickles (magic, protocol, sys_info, storage keys): plain ints / strs / dicts / lists only.
    Resolving ANY global is refused, so none of them can run code either."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError('global {}.{} is not allowed in a model-zoo file'.format(module, name))

    def persistent_load(self, pid):
        raise pickle.UnpicklingError('unexpected persistent id {!r}'.format(pid))


def _plain(f, kind, what):
    v = _PlainUnpickler(f).load()
    if not isinstance(v, kind) or isinstance(v, bool):
        raise ValueError('malformed legacy file: {} is a {}'.format(what, type(v).__name__))
    return v


def _strip(sd):
    """Accept the common wrappings: {'state_dict': ...} and DataParallel's 'module.' prefix
    (python/pytoch_to_onnx.py:14-20 strips the same prefix)."""
    if isinstance(sd, dict) and 'state_dict' in sd and isinstance(sd['state_dict'], dict):
        sd = sd['state_dict']
    out = OrderedDict()
    for k, v in sd.items():
        out[k[7:] if k.startswith('module.') else k] = v
    return out


def load_state_dict_file(path_or_file):
    """Parse a legacy-format zoo file into OrderedDict[name -> fp32 ndarray]."""
    f = open(path_or_file, 'rb') if isinstance(path_or_file, (str, bytes)) or hasattr(path_or_file, '__fspath__') else path_or_file
    try:
        head = f.read(2)
        f.seek(-len(head), io.SEEK_CUR)
        if head == b'PK':
            raise ValueError('zip-format checkpoint: the MoePhoto zoo uses the legacy format; convert it first')
        magic = _plain(f, int, 'the magic number')
        if magic != MAGIC_NUMBER:
            raise ValueError('not a torch legacy-format file (bad magic)')
        proto = _plain(f, int, 'the protocol version')
        if proto != PROTOCOL_VERSION:
            raise ValueError('unsupported legacy protocol {}'.format(proto))
        sys_info = _plain(f, dict, 'sys_info')
        if not sys_info.get('little_endian', True):
            raise ValueError('big-endian checkpoints are not supported')
        up = _ZooUnpickler(f)
        obj = up.load()
        if not isinstance(obj, dict):
            raise ValueError('malformed legacy file: the payload is a {}, not a state dict'.format(type(obj).__name__))
        keys = _plain(f, list, 'the storage key list')
        for key in keys:
            if not isinstance(key, str):
                raise ValueError('malformed legacy file: storage key {!r}'.format(key))
            st = up.storages.get(key)
            head8 = f.read(8)
            if len(head8) != 8:
                raise ValueError('truncated storage header {}'.format(key))
            (numel,) = struct.unpack('<q', head8)
            if numel < 0 or (st is not None and numel != st.numel):
                raise ValueError('storage {} holds {} elements, the state dict announced {}'.format(key, numel, st.numel if st is not None else '?'))
            dt = st.stype.dtype if st is not None else np.dtype('<f4')
            raw = f.read(numel * dt.itemsize)
            if len(raw) != numel * dt.itemsize:
                raise ValueError('truncated storage {}'.format(key))
            if st is not None:
                st.data = np.frombuffer(raw, dtype=dt)
    finally:
        if f is not path_or_file:
            f.close()
    sd = OrderedDict()
    for k, v in _strip(obj).items():
        a = v.materialize() if isinstance(v, _LazyTensor) else np.asarray(v)
        sd[k] = np.array(a, dtype=np.float32) if a.dtype.kind == 'f' else np.array(a)
    return sd


def save_state_dict_file(sd, path):
    """Write `sd` (name -> ndarray) in the legacy layout, byte-compatible with what
    `torch.load(path, weights_only=False)` and the reference's loader expect.  Used to materialise
    synthetic a3/a4/l15/l25/l50 weights (absent from the reference mount) in the zoo's own format."""
    # Hand-assembled protocol-2 pickle: keeps this writer free of any torch import.
    out = io.BytesIO()
    w = out.write

    def binunicode(s):
        b = s.encode('utf-8')
        w(b'X' + struct.pack('<I', len(b)) + b)

    def glob(module, name):
        w(b'c' + module.encode() + b'\n' + name.encode() + b'\n')

    def binint(v):
        if 0 <= v < 256:
            w(b'K' + struct.pack('<B', v))
        elif 0 <= v < 65536:
            w(b'M' + struct.pack('<H', v))
        else:
            w(b'J' + struct.pack('<i', v))

    def int_tuple(t):
        w(b'(')
        for v in t:
            binint(int(v))
        w(b't')

    w(b'\x80\x02')
    glob('collections', 'OrderedDict')
    w(b')R')           # OrderedDict()
    w(b'(')            # MARK for SETITEMS
    arrays = []
    for i, (k, v) in enumerate(sd.items()):
        a = np.ascontiguousarray(np.asarray(v), dtype='<f4')
        arrays.append(a)
        binunicode(k)
        glob('torch._utils', '_rebuild_tensor_v2')
        w(b'(')
        # persistent id ('storage', torch.FloatStorage, key, 'cpu', numel, None)
        w(b'(')
        binunicode('storage')
        glob('torch', 'FloatStorage')
        binunicode(str(i))
        binunicode('cpu')
        binint(a.size)
        w(b'N')
        w(b't')
        w(b'Q')        # BINPERSID
        binint(0)      # storage offset
        int_tuple(a.shape)
        strides = tuple(int(s // 4) for s in a.strides) if a.ndim else ()
        int_tuple(strides)
        w(b'\x89')     # requires_grad False
        glob('collections', 'OrderedDict')
        w(b')R')       # backward hooks
        w(b't')
        w(b'R')        # _rebuild_tensor_v2(*args)
    w(b'u')            # SETITEMS
    w(b'.')
    body = out.getvalue()
    with open(path, 'wb') as f:
        pickle.dump(MAGIC_NUMBER, f, protocol=2)
        pickle.dump(PROTOCOL_VERSION, f, protocol=2)
        pickle.dump({'protocol_version': PROTOCOL_VERSION, 'little_endian': True,
                     'type_sizes': {'short': 2, 'int': 4, 'long': 4}}, f, protocol=2)
        f.write(body)
        pickle.dump([str(i) for i in range(len(arrays))], f, protocol=2)
        for a in arrays:
            f.write(struct.pack('<q', a.size))
            f.write(a.tobytes())
    return path
