#!/usr/bin/env python
"""Writes tests/golden/mxnet_ndarray_list-0003.params BYTE BY BYTE from MXNet's NDArray-list layout, independently of
lsfa_amd/utils/load_model.py's writer: only `struct.pack` calls spelled out below, no numpy .tobytes
of a structure the reader also knows how to write.  Layout per MXNet (src/ndarray/ndarray.cc, NDArray::Save / Load and
mx.nd.save's list container, as of the commit the reference pins, 75a9e187d — un-vendored, so still "unpinned" against
real MXNet output; what this fixture pins is that the reader follows the documented layout rather than its own writer):

  list:      uint64 0x112 | uint64 0 (reserved) | uint64 count | count x NDArray | uint64 n_names | n_names x (uint64 len | bytes)
  NDArray V2: uint32 0xF993FAC9 | int32 storage type (0 = dense) | uint32 ndim | int64 dim[ndim] | int32 dev_type | int32 dev_id
              | int32 type flag (0 = float32, 1 = float64, 4 = int32 ...) | raw little-endian data
  NDArray V1: uint32 0xF993FAC8 | uint32 ndim | int64 dim[ndim] | ctx | type flag | data        (no storage type)
  legacy:     uint32 ndim | uint32 dim[ndim] | ctx | type flag | data                             (no magic, 32-bit dims)

The arrays (expected values are restated in tests/test_load_model_cpu.py):
  arg:conv_weight       V2 float32 (2,3,1,1) = 0.5 * [0..5]
  arg:conv_bias         V2 float32 (3,)      = [-1, 0, 1]
  arg:rfcn_bbox_weight_test   V1 float32 (2,2) = [[1, 2], [3, 4]]       (`process=True` renames it to rfcn_bbox_weight)
  aux:bn_moving_var     legacy float32 (4,)  = [1, 1, 1, 1]
  aux:counter           V2 int32 (2,)        = [7, -7]
"""
import os
import struct

out = []
put = out.append
put(struct.pack('<Q', 0x112))
put(struct.pack('<Q', 0))
put(struct.pack('<Q', 5))
# 1. V2 float32 (2,3,1,1)
put(struct.pack('<I', 0xF993FAC9)); put(struct.pack('<i', 0)); put(struct.pack('<I', 4)); put(struct.pack('<4q', 2, 3, 1, 1))
put(struct.pack('<ii', 1, 0)); put(struct.pack('<i', 0)); put(struct.pack('<6f', 0.0, 0.5, 1.0, 1.5, 2.0, 2.5))
# 2. V2 float32 (3,)
put(struct.pack('<I', 0xF993FAC9)); put(struct.pack('<i', 0)); put(struct.pack('<I', 1)); put(struct.pack('<q', 3))
put(struct.pack('<ii', 1, 0)); put(struct.pack('<i', 0)); put(struct.pack('<3f', -1.0, 0.0, 1.0))
# 3. V1 float32 (2,2), saved from gpu(1)
put(struct.pack('<I', 0xF993FAC8)); put(struct.pack('<I', 2)); put(struct.pack('<2q', 2, 2))
put(struct.pack('<ii', 2, 1)); put(struct.pack('<i', 0)); put(struct.pack('<4f', 1.0, 2.0, 3.0, 4.0))
# 4. legacy float32 (4,)
put(struct.pack('<I', 1)); put(struct.pack('<I', 4))
put(struct.pack('<ii', 1, 0)); put(struct.pack('<i', 0)); put(struct.pack('<4f', 1.0, 1.0, 1.0, 1.0))
# 5. V2 int32 (2,)
put(struct.pack('<I', 0xF993FAC9)); put(struct.pack('<i', 0)); put(struct.pack('<I', 1)); put(struct.pack('<q', 2))
put(struct.pack('<ii', 1, 0)); put(struct.pack('<i', 4)); put(struct.pack('<2i', 7, -7))
names = [b'arg:conv_weight', b'arg:conv_bias', b'arg:rfcn_bbox_weight_test', b'aux:bn_moving_var', b'aux:counter']
put(struct.pack('<Q', len(names)))
for n in names:
    put(struct.pack('<Q', len(n))); put(n)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mxnet_ndarray_list-0003.params')
with open(path, 'wb') as f:
    f.write(b''.join(out))
print(path, sum(len(b) for b in out), 'bytes')
