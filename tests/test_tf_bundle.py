"""TensorFlow V2 checkpoint bundles without TensorFlow (kpx_amd.tf_bundle).  PARITY UNPINNED against TensorFlow
itself (none here): the checks are published CRC-32C vectors, the protobuf layer against google.protobuf with the public field
numbers of tensor_bundle.proto / tensor_shape.proto, table invariants and write -> read round trips."""
import os
import struct

import numpy as np
import pytest


@pytest.fixture(scope='module')
def tb():
    import kpx_amd.tf_bundle as m           # needs libkpx_hip.so for the CRC (host code; no GPU)
    return m


def test_crc32c_known_answers_and_masking(tb):
    assert tb.crc32c(b'123456789') == 0xE3069283
    assert tb.crc32c(bytes(32)) == 0x8A9136AA and tb.crc32c(b'\xff' * 32) == 0x62A8AB43          # RFC 3720 B.4
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E and tb.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    a = np.random.RandomState(0).bytes(100003)
    assert tb.crc32c(a[777:], tb.crc32c(a[:777])) == tb.crc32c(a)
    for c in (0, 1, 0xE3069283, 0xffffffff):
        assert tb.unmask_crc(tb.mask_crc(c)) == c
    assert tb.mask_crc(0) == 0xa282ead8


def _proto_classes():
    """BundleEntryProto / TensorShapeProto / BundleHeaderProto rebuilt from their public field numbers with google.protobuf."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='kpx_bundle_test.proto', package='kpxt', syntax='proto3')
    dim = descriptor_pb2.DescriptorProto(name='Dim')
    dim.field.add(name='size', number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    dim.field.add(name='name', number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    shape = fd.message_type.add(name='TensorShapeProto')
    shape.nested_type.append(dim)
    shape.field.add(name='dim', number=2, type=F.TYPE_MESSAGE, type_name='.kpxt.TensorShapeProto.Dim', label=F.LABEL_REPEATED)
    shape.field.add(name='unknown_rank', number=3, type=F.TYPE_BOOL, label=F.LABEL_OPTIONAL)
    ent = fd.message_type.add(name='BundleEntryProto')
    ent.field.add(name='dtype', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ent.field.add(name='shape', number=2, type=F.TYPE_MESSAGE, type_name='.kpxt.TensorShapeProto', label=F.LABEL_OPTIONAL)
    ent.field.add(name='shard_id', number=3, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ent.field.add(name='offset', number=4, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name='size', number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name='crc32c', number=6, type=F.TYPE_FIXED32, label=F.LABEL_OPTIONAL)
    ver = fd.message_type.add(name='VersionDef')
    ver.field.add(name='producer', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ver.field.add(name='min_consumer', number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    hdr = fd.message_type.add(name='BundleHeaderProto')
    hdr.field.add(name='num_shards', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    hdr.field.add(name='endianness', number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    hdr.field.add(name='version', number=3, type=F.TYPE_MESSAGE, type_name='.kpxt.VersionDef', label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('kpxt.' + n))
    return get('BundleEntryProto'), get('BundleHeaderProto')


def test_protobuf_fragments_match_google_protobuf(tb):
    Entry, Header = _proto_classes()
    h = Header(num_shards=1); h.version.producer = 1
    assert tb.encode_header() == h.SerializeToString()
    for shape, off, size, crc in (((3, 3, 64, 128), 0, 294912, 0x12345678), ((), 1234567890123, 4, 0xffffffff), ((2048,), 77, 8192, 1)):
        e = Entry(dtype=1, offset=off, size=size, crc32c=crc)
        e.shape.SetInParent()
        for s in shape:
            e.shape.dim.add(size=s)
        mine = tb.encode_entry(1, shape, off, size, crc)
        assert mine == e.SerializeToString(), (shape, mine.hex(), e.SerializeToString().hex())
        d = tb.decode_entry(mine)
        assert (d['dtype'], tuple(d['shape']), d['offset'], d['size'], d['crc32c']) == (1, tuple(shape), off, size, crc)
    # the example worked out by hand in DESIGN.md: float [3,3,64,128] at offset 0, 294912 bytes
    assert tb.encode_entry(1, (3, 3, 64, 128), 0, 294912, 0x12345678).hex() == '08011211120208031202080312020840120308800128808012' + '3578563412'


def test_table_layout_invariants_and_round_trip(tb, tmp_path, monkeypatch):
    monkeypatch.setattr(tb, 'BLOCK_SIZE', 512)                 # force many data blocks
    items = [(b'', b'header')] + [(('scope_%03d/conv_%d/kernel' % (i // 7, i % 7)).encode(), os.urandom(20 + i % 13)) for i in range(300)]
    items.sort()
    path = str(tmp_path / 't.index')
    tb.write_table(path, items)
    buf = open(path, 'rb').read()
    assert struct.unpack_from('<Q', buf, len(buf) - 8)[0] == 0xdb4775248b80fb57 and len(buf) > 48
    assert tb.read_table(path) == items
    # first data block starts at offset 0 with the empty key: shared = 0, non_shared = 0, value_length = 6, then the value
    assert buf[:9] == b'\x00\x00\x06header'
    bad = bytearray(buf); bad[30] ^= 0x40
    open(path, 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        tb.read_table(path)


def test_bundle_round_trip_partial_restore_and_corruption(tb, tmp_path):
    rs = np.random.RandomState(3)
    arrays = {'img_discr/conv_0/kernel': rs.randn(4, 4, 3, 64).astype(np.float32),
              'img_discr/conv_0/kernel/Adam': rs.randn(4, 4, 3, 64).astype(np.float32),
              'pose_encoder/encoder/batch_norm_1/moving_variance': rs.rand(32).astype(np.float32),
              'beta1_power': np.float32(0.5 ** 7), 'global_step': np.int64(123456),
              'vae_decoder/multi_rnn_cell/cell_0/basic_lstm_cell/kernel': rs.randn(96, 256).astype(np.float32)}
    prefix = str(tmp_path / 'ck' / 'model.ckpt-7')
    tb.write_bundle(prefix, arrays)
    assert tb.is_bundle(prefix) and os.path.getsize(prefix + '.data-00000-of-00001') == sum(np.asarray(a).nbytes for a in arrays.values())
    assert open(str(tmp_path / 'ck' / 'checkpoint')).read().startswith('model_checkpoint_path: "model.ckpt-7"')
    got = tb.read_bundle(prefix)
    assert sorted(got) == sorted(arrays)
    for k, a in arrays.items():
        assert got[k].dtype == np.asarray(a).dtype and got[k].shape == np.asarray(a).shape and np.array_equal(got[k], a), k
    assert tb.list_bundle(prefix)['img_discr/conv_0/kernel'] == (np.float32, (4, 4, 3, 64)) and tb.list_bundle(prefix)['global_step'] == (np.int64, ())
    part = tb.read_bundle(prefix, names={'beta1_power', 'not_there'})
    assert list(part) == ['beta1_power'] and part['beta1_power'] == np.float32(0.5 ** 7)
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read()); data[100] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    with pytest.raises(ValueError, match='checksum'):
        tb.read_bundle(prefix)


def test_bundle_assembled_from_the_public_spec_is_read_and_reproduced_bit_for_bit(tb, tmp_path, golden_dir):
    """tests/golden/tiny_bundle.* was assembled by tests/golden/make_bundle_golden.py from the public format definitions with code
    that shares nothing with tf_bundle.py (google.protobuf messages, its own bit-wise CRC-32C and table builder).  No TensorFlow-written
    file exists in this environment; this pins the reader / writer to a second implementation of the published format."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_bundle_golden', os.path.join(golden_dir, 'make_bundle_golden.py'))
    gold = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gold)
    want = gold.tensors()
    prefix = os.path.join(golden_dir, 'tiny_bundle')
    listed = tb.list_bundle(prefix)
    assert sorted(listed) == sorted(want)
    assert listed['global_step'] == (np.int32, ()) and listed['img_discr/D_logit/conv2d/kernel'] == (np.float32, (3, 3, 4, 1))
    got = tb.read_bundle(prefix)
    for name, a in want.items():
        assert got[name].dtype == np.asarray(a).dtype and got[name].shape == np.asarray(a).shape, name
        assert np.array_equal(got[name], np.asarray(a)), name
    out = os.path.join(str(tmp_path), 'tiny_bundle')
    tb.write_bundle(out, want)
    for ext in ('.index', '.data-00000-of-00001'):
        assert open(out + ext, 'rb').read() == open(prefix + ext, 'rb').read(), ext
    # independent CRC agrees with the C library's slice-by-8 on every tensor
    for a in want.values():
        raw = np.asarray(a).tobytes()
        assert gold.crc32c_bitwise(raw) == tb.crc32c(raw)
