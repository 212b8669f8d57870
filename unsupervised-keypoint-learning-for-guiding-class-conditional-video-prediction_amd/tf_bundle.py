"""TensorFlow V2 checkpoint bundles without TensorFlow.

The reference saves / restores through ``tf.train.Saver`` (models/base_model.py:74-91): ``<prefix>.index`` +
``<prefix>.data-00000-of-00001`` (+ the text file ``checkpoint``).  This module reads and writes that container for the
variable manifest of SURVEY Appendix B, so the published stage-1 / stage-2 checkpoints can be restored here and the reference
can restore checkpoints written here.  Formats restated from their public definitions:

* ``.data-00000-of-00001``: the tensors' raw little-endian bytes back to back, in key order (tensor_bundle.cc BundleWriter::Add);
* ``.index``: a LevelDB-format table (tensorflow/core/lib/io/format.cc, table_builder.cc; magic 0xdb4775248b80fb57):
  data blocks of prefix-compressed ``key -> value`` entries with a restart array, each followed by a 5-byte trailer
  (compression type, masked CRC-32C), an index block of ``last key -> BlockHandle``, an empty metaindex block and a 48-byte
  footer.  Key "" maps to a ``BundleHeaderProto`` (num_shards = 1, little endian, version producer 1), every other key is a
  variable name mapping to a ``BundleEntryProto`` (dtype, shape, shard_id, offset, size, masked CRC-32C of the bytes);
* masked CRC: ``((crc >> 15) | (crc << 17)) + 0xa282ead8`` (lib/hash/crc32c.h).

PARITY UNPINNED against TensorFlow itself: neither TensorFlow nor a TensorFlow-written checkpoint exists in this environment.
What is checked (tests/test_tf_bundle.py): the published CRC-32C test vectors; the protobuf layer against google.protobuf; and a
complete bundle assembled byte by byte from the public format definitions by an independent second implementation
(tests/golden/make_bundle_golden.py: its own bit-wise CRC-32C, google.protobuf messages, its own table builder), which ``read_bundle``
must parse and ``write_bundle`` must reproduce bit for bit.  The reader verifies every checksum it meets and refuses compressed
blocks loudly, so a real bundle that parses here parsed correctly.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
BLOCK_SIZE = 256 * 1024           # BundleWriter's table options; any size is readable
RESTART_INTERVAL = 16

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}


def crc32c(data, crc=0):
    from ._lib import lib
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    return int(lib.kpx_crc32c_host(crc, buf, len(buf)))


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------------------------ varints / protobuf fragments
def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _pb_fields(buf):
    """-> list of (field number, wire type, value); value = int for varint / fixed, bytes for length-delimited."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((field, wt, v))
    return out


def encode_header():
    """BundleHeaderProto{num_shards: 1, endianness: LITTLE (0, default), version: VersionDef{producer: 1}}"""
    return b'\x08\x01' + b'\x1a\x02\x08\x01'


def encode_entry(dtype_id, shape, offset, size, masked_crc):
    """BundleEntryProto{dtype=1, shape=2{dim=2{size=1}}, shard_id=3 (0: omitted), offset=4, size=5, crc32c=6 fixed32}"""
    dims = b''.join(b'\x12' + _varint(len(d)) + d for d in (b'\x08' + _varint(int(s)) for s in shape))
    out = b'\x08' + _varint(dtype_id) + b'\x12' + _varint(len(dims)) + dims
    if offset:
        out += b'\x20' + _varint(offset)
    out += b'\x28' + _varint(size) + b'\x35' + struct.pack('<I', masked_crc)
    return out


def decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for field, wt, v in _pb_fields(buf):
        if field == 1:
            e['dtype'] = v
        elif field == 2:
            for f2, _, dim in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, dv in _pb_fields(dim):
                        if f3 == 1:
                            size = dv - (1 << 64) if dv >> 63 else dv
                    e['shape'].append(size)
        elif field == 3:
            e['shard_id'] = v
        elif field == 4:
            e['offset'] = v
        elif field == 5:
            e['size'] = v
        elif field == 6:
            e['crc32c'] = v
        elif field == 7:
            e['slices'] += 1
    return e


# ------------------------------------------------------------------------------------------------ LevelDB-format table
def _build_block(items, restart_interval):
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def _parse_block(block):
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen]))); pos += vlen
    return out


def _emit_block(f, content):
    """block + trailer (type 0 = uncompressed, masked crc32c of content + type) -> (offset, size) handle"""
    offset = f.tell()
    f.write(content)
    f.write(b'\x00' + struct.pack('<I', mask_crc(crc32c(content + b'\x00'))))
    return offset, len(content)


def _read_block(buf, offset, size):
    content, ctype = buf[offset:offset + size], buf[offset + size]
    stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
    if unmask_crc(stored) != crc32c(bytes(content) + bytes([ctype])):
        raise ValueError('table block checksum mismatch at offset %d' % offset)
    if ctype != 0:
        raise ValueError('compressed table block (type %d): tensor bundles are written uncompressed' % ctype)
    return content


def _short_separator(start, limit):
    """leveldb BytewiseComparator::FindShortestSeparator: a short key k with start <= k < limit."""
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xff and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _short_successor(key):
    """leveldb BytewiseComparator::FindShortSuccessor: the first byte that is not 0xff incremented, the rest dropped."""
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def write_table(path, items):
    """items: sorted list of (key bytes, value bytes).  Follows TableBuilder (core/lib/io/table_builder.cc = LevelDB's): a data block is
    flushed once its encoded size (entries + restart array + count) reaches BLOCK_SIZE; the index key of a block is the shortest
    separator between its last key and the next block's first key, the short successor of the last key for the final block."""
    with open(path, 'wb') as f:
        index, block, pending = [], [], None            # pending = (last key of the flushed block, its handle)
        est = 0                                         # running BlockBuilder::CurrentSizeEstimate(): entry bytes so far (restart array added below)
        for k, v in items:
            if pending is not None:
                index.append((_short_separator(pending[0], k), pending[1]))
                pending = None
            # size of this entry as _build_block will encode it (shared-prefix compression against the previous key, none at a restart point)
            shared = 0
            if len(block) % RESTART_INTERVAL != 0:
                last = block[-1][0]
                while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                    shared += 1
            est += len(_varint(shared)) + len(_varint(len(k) - shared)) + len(_varint(len(v))) + (len(k) - shared) + len(v)
            block.append((k, v))
            n_restarts = (len(block) + RESTART_INTERVAL - 1) // RESTART_INTERVAL
            if est + 4 * n_restarts + 4 >= BLOCK_SIZE:                  # = len(_build_block(block)) without re-encoding the block per key
                h = _emit_block(f, _build_block(block, RESTART_INTERVAL))
                pending = (block[-1][0], _varint(h[0]) + _varint(h[1]))
                block, est = [], 0
        if block:
            h = _emit_block(f, _build_block(block, RESTART_INTERVAL))
            pending = (block[-1][0], _varint(h[0]) + _varint(h[1]))
        if pending is not None:
            index.append((_short_successor(pending[0]), pending[1]))
        meta = _emit_block(f, _build_block([], RESTART_INTERVAL))
        idx = _emit_block(f, _build_block(index, 1))
        footer = _varint(meta[0]) + _varint(meta[1]) + _varint(idx[0]) + _varint(idx[1])
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))


def read_table(path):
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError('%s is not a TensorFlow / LevelDB table (bad magic)' % path)
    pos = len(buf) - 48
    _, pos = _read_varint(buf, pos); _, pos = _read_varint(buf, pos)               # metaindex handle
    ioff, pos = _read_varint(buf, pos); isize, pos = _read_varint(buf, pos)
    items = []
    for _, handle in _parse_block(_read_block(buf, ioff, isize)):
        off, p = _read_varint(handle, 0); sz, _ = _read_varint(handle, p)
        items.extend(_parse_block(_read_block(buf, off, sz)))
    return items


# ------------------------------------------------------------------------------------------------ bundles
def write_bundle(prefix, arrays):
    """arrays: {variable name: numpy array}.  Writes <prefix>.index, <prefix>.data-00000-of-00001 and the ``checkpoint`` state file."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or '.', exist_ok=True)
    items, offset = [(b'', encode_header())], 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(arrays):                       # BundleWriter keeps a sorted map; data follows the same order
            a = np.asarray(arrays[name])               # (ascontiguousarray would turn a scalar into shape [1])
            if a.dtype.byteorder == '>':
                a = a.astype(a.dtype.newbyteorder('<'))
            if a.dtype not in DTYPE_IDS:
                raise ValueError('variable %s: dtype %s has no TensorFlow bundle mapping here' % (name, a.dtype))
            raw = a.tobytes()                          # C order
            f.write(raw)
            items.append((name.encode(), encode_entry(DTYPE_IDS[a.dtype], a.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', items)
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), 'checkpoint'), 'w') as f:      # CheckpointState text proto
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def is_bundle(prefix):
    return os.path.exists(prefix + '.index')


def read_bundle(prefix, names=None):
    """-> {variable name: numpy array}; ``names`` restricts the tensors that are loaded (partial restore, base_model.py:84-91)."""
    items = read_table(prefix + '.index')
    if not items or items[0][0] != b'':
        raise ValueError('%s.index has no bundle header entry' % prefix)
    num_shards, endianness = 1, 0
    for field, _, v in _pb_fields(items[0][1]):
        if field == 1:
            num_shards = v
        elif field == 2:
            endianness = v
    if endianness != 0:
        raise ValueError('big-endian bundle')
    shards, out = {}, {}
    for key, value in items[1:]:
        name = key.decode()
        if names is not None and name not in names:
            continue
        e = decode_entry(value)
        if e['slices']:
            raise ValueError('variable %s is stored as slices of a partitioned variable: not supported' % name)
        if e['dtype'] not in DTYPES:
            raise ValueError('variable %s has unsupported dtype id %d' % (name, e['dtype']))
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), 'rb')
        f = shards[sid]
        f.seek(e['offset'])
        raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise ValueError('variable %s: data file truncated' % name)
        if e['crc32c'] is not None and unmask_crc(e['crc32c']) != crc32c(raw):
            raise ValueError('variable %s: tensor checksum mismatch' % name)
        out[name] = np.frombuffer(raw, dtype=np.dtype(DTYPES[e['dtype']]).newbyteorder('<')).reshape(e['shape']).copy()
    for f in shards.values():
        f.close()
    return out


def list_bundle(prefix):
    """-> {name: (numpy dtype, shape)} without touching the data file"""
    out = {}
    for key, value in read_table(prefix + '.index')[1:]:
        e = decode_entry(value)
        out[key.decode()] = (DTYPES.get(e['dtype']), tuple(e['shape']))
    return out
