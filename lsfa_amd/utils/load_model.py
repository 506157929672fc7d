"""MXNet `.params` checkpoints -> numpy dicts.

API of lib/utils/load_model.py: `load_checkpoint(prefix, epoch)` (:11-31) and
`load_param(prefix, epoch, convert, ctx, process)` (:46-66; `process=True` renames the
`*_test` parameters that callback.do_checkpoint stores with BBOX_STDS folded in, core/callback.py:54-64).

The file format is MXNet's NDArray-list serialisation [MXNet@75a9e187d src/ndarray/ndarray.cc,
un-vendored — restated from the published format, PARITY UNPINNED: the reference ships no .params]:
    uint64 0x112 | uint64 0 | uint64 n | n x NDArray | uint64 n_names | n_names x (uint64 len, bytes)
    NDArray (V2, 0xF993FAC9): uint32 magic | int32 stype(0 = dense) | shape | ctx(int32 dev_type, int32 dev_id)
                              | int32 type_flag | raw data
    NDArray (V1, 0xF993FAC8): uint32 magic | shape | ctx | int32 type_flag | raw data
    NDArray (legacy):         uint32 ndim | uint32 dims[ndim] | ctx | int32 type_flag | raw data
    shape (V1/V2): uint32 ndim | int64 dims[ndim]
"""
import struct

import numpy as np

_LIST_MAGIC = 0x112
_V1, _V2 = 0xF993FAC8, 0xF993FAC9
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


class _Reader(object):
    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def read(self, fmt):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.buf):
            raise ValueError("truncated .params file")
        v = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += size
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if self.pos + n > len(self.buf):
            raise ValueError("truncated .params file")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b


def _read_ndarray(r):
    first = r.read('<I')
    if first == _V2:
        stype = r.read('<i')
        if stype != 0:
            raise ValueError("sparse NDArray (stype %d) is not supported" % stype)
        ndim = r.read('<I')
        shape = tuple(r.read('<%dq' % ndim)) if ndim > 1 else ((r.read('<q'),) if ndim == 1 else ())
    elif first == _V1:
        ndim = r.read('<I')
        shape = tuple(r.read('<%dq' % ndim)) if ndim > 1 else ((r.read('<q'),) if ndim == 1 else ())
    else:                       # legacy: `first` is ndim, dims are uint32
        ndim = first
        shape = tuple(r.read('<%dI' % ndim)) if ndim > 1 else ((r.read('<I'),) if ndim == 1 else ())
    if ndim == 0:
        return None             # is_none()
    r.read('<ii')               # context (dev_type, dev_id)
    flag = r.read('<i')
    if flag not in _DTYPES:
        raise ValueError("unknown NDArray type flag %d" % flag)
    dt = np.dtype(_DTYPES[flag])
    n = int(np.prod(shape))
    return np.frombuffer(r.raw(n * dt.itemsize), dtype=dt).reshape(shape).copy()


def load_ndarray_list(path):
    with open(path, 'rb') as f:
        r = _Reader(f.read())
    if r.read('<Q') != _LIST_MAGIC:
        raise ValueError("%s is not an MXNet NDArray list" % path)
    r.read('<Q')
    n = r.read('<Q')
    arrays = [_read_ndarray(r) for _ in range(n)]
    n_names = r.read('<Q')
    names = [r.raw(r.read('<Q')).decode() for _ in range(n_names)]
    if names and len(names) != len(arrays):
        raise ValueError("name/array count mismatch")
    return dict(zip(names, arrays)) if names else arrays


def save_ndarray_list(path, named):
    """Writes the V2 dense format (what MXNet >= 0.11 writes)."""
    out = [struct.pack('<QQQ', _LIST_MAGIC, 0, len(named))]
    for _, a in named.items():
        a = np.ascontiguousarray(a)
        out.append(struct.pack('<Ii', _V2, 0))
        out.append(struct.pack('<I', a.ndim) + struct.pack('<%dq' % a.ndim, *a.shape))
        out.append(struct.pack('<ii', 1, 0))                      # cpu(0)
        out.append(struct.pack('<i', _FLAGS[a.dtype]))
        out.append(a.tobytes())
    out.append(struct.pack('<Q', len(named)))
    for k in named:
        b = k.encode()
        out.append(struct.pack('<Q', len(b)) + b)
    with open(path, 'wb') as f:
        f.write(b''.join(out))


def load_checkpoint(prefix, epoch):
    """-> (arg_params, aux_params): keys 'arg:name' / 'aux:name' split as in load_model.py:22-31."""
    save_dict = load_ndarray_list('%s-%04d.params' % (prefix, epoch))
    arg_params, aux_params = {}, {}
    for k, v in save_dict.items():
        tp, name = k.split(':', 1)
        if tp == 'arg':
            arg_params[name] = v
        if tp == 'aux':
            aux_params[name] = v
    return arg_params, aux_params


def save_checkpoint(prefix, epoch, arg_params, aux_params):
    """lib/utils/save_model.py:22-25."""
    d = {('arg:%s' % k): v for k, v in arg_params.items()}
    d.update({('aux:%s' % k): v for k, v in aux_params.items()})
    save_ndarray_list('%s-%04d.params' % (prefix, epoch), d)


def load_param(prefix, epoch, convert=False, ctx=None, process=False):
    arg_params, aux_params = load_checkpoint(prefix, epoch)
    if process:
        tests = [k for k in arg_params.keys() if '_test' in k]
        for test in tests:
            arg_params[test.replace('_test', '')] = arg_params.pop(test)
    return arg_params, aux_params
