"""MXNet .params reader (restated format, parity unpinned): round trip, the *_test renaming of
load_param(process=True), hand-assembled legacy / V1 records, and feeding a checkpoint to init_weight."""
import struct

import numpy as np
import pytest

from lsfa_amd.utils import load_model as lm


def test_round_trip_and_process(tmp_path):
    rs = np.random.RandomState(0)
    arg = {'rfcn_bbox_weight': rs.randn(392, 512, 1, 1).astype(np.float32),
           'rfcn_bbox_weight_test': rs.randn(392, 512, 1, 1).astype(np.float32),
           'rfcn_bbox_bias_test': rs.randn(392).astype(np.float32), 'conv0_weight': rs.randn(64, 3, 7, 7).astype(np.float32)}
    aux = {'bn0_moving_mean': rs.randn(64).astype(np.float32), 'bn0_moving_var': rs.rand(64).astype(np.float32)}
    prefix = str(tmp_path / 'lsfa')
    lm.save_checkpoint(prefix, 2, arg, aux)
    a2, x2 = lm.load_checkpoint(prefix, 2)
    assert set(a2) == set(arg) and set(x2) == set(aux)
    for k in arg:
        np.testing.assert_array_equal(a2[k], arg[k])
    a3, _ = lm.load_param(prefix, 2, process=True)
    np.testing.assert_array_equal(a3['rfcn_bbox_weight'], arg['rfcn_bbox_weight_test'])   # load_model.py:62-65
    assert 'rfcn_bbox_weight_test' not in a3 and 'rfcn_bbox_bias' in a3


def test_legacy_and_v1_records(tmp_path):
    data = np.arange(6, dtype=np.float32).reshape(2, 3)
    legacy = struct.pack('<I2I', 2, 2, 3) + struct.pack('<ii', 1, 0) + struct.pack('<i', 0) + data.tobytes()
    v1 = struct.pack('<II2q', 0xF993FAC8, 2, 2, 3) + struct.pack('<ii', 2, 0) + struct.pack('<i', 0) + data.tobytes()
    names = b''.join(struct.pack('<Q', len(n)) + n for n in (b'arg:a', b'aux:b'))
    blob = struct.pack('<QQQ', 0x112, 0, 2) + legacy + v1 + struct.pack('<Q', 2) + names
    p = tmp_path / 'x-0000.params'
    p.write_bytes(blob)
    arg, aux = lm.load_checkpoint(str(tmp_path / 'x'), 0)
    np.testing.assert_array_equal(arg['a'], data)
    np.testing.assert_array_equal(aux['b'], data)
    p.write_bytes(blob[:40])
    with pytest.raises(ValueError):
        lm.load_checkpoint(str(tmp_path / 'x'), 0)


def test_checkpoint_feeds_init_weight(tmp_path):
    """A backbone-only checkpoint + init_weight gives a complete parameter set (:753-870)."""
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.symbols import params as P
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg = lsfa_test_config(10)
    full_arg, full_aux = P.init_params(cfg, seed=9)
    keep = lambda k: not k.startswith(('small_net_', 'Nq_', 'rnet_', 'fuse_reduce_add', 'rfcn_', 'rpn_'))
    small = {k: v for k, v in full_arg.items() if k.startswith(('conv0', 'bn0', 'bn_data', 'stage1_unit1_bn1'))}
    prefix = str(tmp_path / 'pre')
    lm.save_checkpoint(prefix, 0, small, {k: v for k, v in full_aux.items() if k.startswith(('bn0', 'bn_data', 'stage1_unit1_bn1'))})
    arg, aux = lm.load_param(prefix, 0, process=True)
    net = resnet_v1_101_flownet_rfcn(cfg)
    sym = net.get_cur_test_symbol(cfg)
    net.init_weight(cfg, arg, aux, seed=9)
    assert set(sym.arg_spec) <= set(arg) and set(sym.aux_spec) <= set(aux)
    np.testing.assert_array_equal(arg['small_net_conv0_weight'], small['conv0_weight'])    # copied from the big net
    assert keep('conv0_weight')


def test_fixture_written_byte_by_byte_from_the_documented_layout():
    """tests/golden/mxnet_ndarray_list-0003.params was assembled with struct.pack calls only (make_params_fixture.py), not by
    this package's writer: V2, V1 and legacy NDArray records, float32 and int32, a record saved from a GPU context, and the
    `_test` renaming of load_param(process=True) (lib/utils/load_model.py:57-62)."""
    import os
    import numpy as np
    from lsfa_amd.utils.load_model import load_param, load_ndarray_list
    prefix = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mxnet_ndarray_list')
    raw = load_ndarray_list(prefix + '-0003.params')
    assert list(raw) == ['arg:conv_weight', 'arg:conv_bias', 'arg:rfcn_bbox_weight_test', 'aux:bn_moving_var', 'aux:counter']
    arg, aux = load_param(prefix, 3, process=True)
    assert sorted(arg) == ['conv_bias', 'conv_weight', 'rfcn_bbox_weight'] and sorted(aux) == ['bn_moving_var', 'counter']
    np.testing.assert_array_equal(arg['conv_weight'], (0.5 * np.arange(6, dtype=np.float32)).reshape(2, 3, 1, 1))
    assert arg['conv_weight'].dtype == np.float32
    np.testing.assert_array_equal(arg['conv_bias'], np.array([-1, 0, 1], np.float32))
    np.testing.assert_array_equal(arg['rfcn_bbox_weight'], np.array([[1, 2], [3, 4]], np.float32))
    np.testing.assert_array_equal(aux['bn_moving_var'], np.ones(4, np.float32))
    np.testing.assert_array_equal(aux['counter'], np.array([7, -7], np.int32))
    assert aux['counter'].dtype == np.int32
