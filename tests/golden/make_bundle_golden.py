#!/usr/bin/env python3
"""Generate tests/golden/tiny_bundle.{index,data-00000-of-00001}: a TensorFlow V2 checkpoint bundle assembled byte by byte from the
PUBLIC format definitions, by code that shares nothing with kpx_amd/tf_bundle.py:

    python -B tests/golden/make_bundle_golden.py

* protobuf values come from google.protobuf messages rebuilt from the public field numbers of tensor_bundle.proto
  (BundleHeaderProto, BundleEntryProto), tensor_shape.proto (TensorShapeProto) and versions.proto (VersionDef);
* the .index file follows LevelDB's table_format.md as TensorFlow's core/lib/io/table_builder.cc implements it: prefix-compressed
  entries (shared | non_shared | value_length varints), a restart point every 16 entries, the restart array and its count, a 5-byte
  block trailer (type 0 = uncompressed, masked CRC-32C of block + type), an empty metaindex block, an index block (restart
  interval 1) whose key for the last data block is the SHORT SUCCESSOR of that block's last key (BytewiseComparator), and the
  48-byte footer (two BlockHandles padded to 40 bytes + magic 0xdb4775248b80fb57 little endian);
* CRC-32C is a bit-at-a-time Castagnoli implementation written here (polynomial 0x82F63B78 reflected), masked as
  ((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32;
* the .data file is the tensors' little-endian bytes back to back in key order.

This is NOT a TensorFlow-written file -- none exists in this environment, and TensorFlow cannot be installed -- so it pins
tf_bundle.py to the published format as read by a second, independent implementation, not to TensorFlow's own output.
The tensors are regenerated from seeds by tests/test_tf_bundle.py.
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def tensors():
    rs = np.random.RandomState(20260101)
    t = {'global_step': np.int32(12345),                                                 # reference train.py:30
         'beta1_power': np.float32(0.5) ** 3, 'beta2_power_1': np.float32(0.999) ** 3,
         'img_discr/D_logit/conv2d/kernel': rs.randn(3, 3, 4, 1).astype(np.float32),
         'img_discr/D_logit/conv2d/kernel/Adam': rs.randn(3, 3, 4, 1).astype(np.float32),
         'pose_encoder/conv_0/conv2d/bias': rs.randn(3).astype(np.float32),
         'translator/b_norm_1_0/moving_variance': rs.rand(8).astype(np.float32) + 0.5}
    for i in range(20):                                                                  # more than one restart interval
        t['vae_decoder/multi_rnn_cell/cell_%d/basic_lstm_cell/w%02d' % (i % 2, i)] = rs.randn(2, i % 3 + 1).astype(np.float32)
    return t


def crc32c_bitwise(data):
    crc = 0xffffffff
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xffffffff


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def varint(v):
    out = b''
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def proto_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='kpx_bundle_golden.proto', package='kpxg', syntax='proto3')
    shape = fd.message_type.add(name='TensorShapeProto')
    dim = shape.nested_type.add(name='Dim')
    dim.field.add(name='size', number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    dim.field.add(name='name', number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    shape.field.add(name='dim', number=2, type=F.TYPE_MESSAGE, type_name='.kpxg.TensorShapeProto.Dim', label=F.LABEL_REPEATED)
    shape.field.add(name='unknown_rank', number=3, type=F.TYPE_BOOL, label=F.LABEL_OPTIONAL)
    ver = fd.message_type.add(name='VersionDef')
    ver.field.add(name='producer', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ver.field.add(name='min_consumer', number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    hdr = fd.message_type.add(name='BundleHeaderProto')
    hdr.field.add(name='num_shards', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    hdr.field.add(name='endianness', number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)          # enum LITTLE = 0
    hdr.field.add(name='version', number=3, type=F.TYPE_MESSAGE, type_name='.kpxg.VersionDef', label=F.LABEL_OPTIONAL)
    ent = fd.message_type.add(name='BundleEntryProto')
    ent.field.add(name='dtype', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)               # enum DataType
    ent.field.add(name='shape', number=2, type=F.TYPE_MESSAGE, type_name='.kpxg.TensorShapeProto', label=F.LABEL_OPTIONAL)
    ent.field.add(name='shard_id', number=3, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ent.field.add(name='offset', number=4, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name='size', number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name='crc32c', number=6, type=F.TYPE_FIXED32, label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('kpxg.' + n))
    return get('BundleHeaderProto'), get('BundleEntryProto')


DT = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9}      # types.proto: DT_FLOAT, DT_INT32, DT_INT64


def block(items, restart_interval):
    out, restarts, last = b'', [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < len(last) and shared < len(k) and last[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    for r in (restarts or [0]):
        out += struct.pack('<I', r)
    return out + struct.pack('<I', len(restarts or [0]))


def short_successor(key):
    """leveldb BytewiseComparatorImpl::FindShortSuccessor: first byte that is not 0xff, incremented; the rest dropped."""
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def main():
    Header, Entry = proto_classes()
    t = tensors()
    data, items = b'', []
    h = Header(num_shards=1)
    h.version.producer = 1
    items.append((b'', h.SerializeToString()))
    for name in sorted(t):
        a = np.asarray(t[name])
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        e = Entry(dtype=DT[a.dtype], offset=len(data), size=len(raw), crc32c=masked(crc32c_bitwise(raw)))
        e.shape.SetInParent()
        for s in a.shape:
            e.shape.dim.add(size=int(s))
        items.append((name.encode(), e.SerializeToString()))
        data += raw
    with open(os.path.join(HERE, 'tiny_bundle.data-00000-of-00001'), 'wb') as f:
        f.write(data)
    out = b''

    def emit(content):
        nonlocal out
        handle = varint(len(out)) + varint(len(content))
        out += content + b'\x00' + struct.pack('<I', masked(crc32c_bitwise(content + b'\x00')))
        return handle
    data_handle = emit(block(items, 16))                      # everything fits one 256 KB data block
    meta_handle = emit(block([], 16))
    index_handle = emit(block([(short_successor(items[-1][0]), data_handle)], 1))
    footer = meta_handle + index_handle
    out += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    with open(os.path.join(HERE, 'tiny_bundle.index'), 'wb') as f:
        f.write(out)
    print('wrote tiny_bundle.index (%d B), tiny_bundle.data-00000-of-00001 (%d B), %d tensors' % (len(out), len(data), len(t)))


if __name__ == '__main__':
    main()
