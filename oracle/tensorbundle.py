"""TEST INFRASTRUCTURE ONLY -- not part of the product path.

Read-only parser for TensorFlow "TensorBundle" checkpoints (``ckpt-N.index`` +
``ckpt-N.data-00000-of-00001``).  It executes nothing from the files: the index is
a LevelDB-format sorted table of (key -> BundleEntryProto) and the data shard is raw
little-endian tensor bytes.  Used only by ``scripts/make_golden_dssm.py`` to turn the
reference's committed artifact ``2.FM/retrieval_model/checkpoint/ckpt-7.*`` into the
small fixtures under ``tests/golden/``.

Format notes (public TensorFlow / LevelDB on-disk formats):
  * footer (last 48 bytes): metaindex BlockHandle, index BlockHandle (varint64 offset,
    varint64 size each), zero padding, 8-byte magic 0xdb4775248b80fb57 (little endian);
  * a block = entries + uint32 restart offsets + uint32 n_restarts, followed on disk by
    a 5-byte trailer (1 byte compression type, 4 byte crc);
  * an entry = varint32 shared, varint32 non_shared, varint32 value_len, key delta, value;
  * BundleEntryProto fields: 1 dtype, 2 shape (TensorShapeProto: repeated dim{1:size}),
    3 shard_id, 4 offset, 5 size, 6 crc32c.
"""
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}


def _varint(buf, pos):
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block_entries(buf, offset, size):
    """Yield (key, value) of one uncompressed table block."""
    if buf[offset + size] != 0:
        raise ValueError("compressed table block (type %d) not supported" % buf[offset + size])
    blk = buf[offset:offset + size]
    n_restarts = struct.unpack_from("<I", blk, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        yield key, blk[pos:pos + vlen]
        pos += vlen


def _proto_fields(buf):
    """Minimal protobuf wire decoder: yields (field_no, wire_type, value)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield field, wt, val


def _parse_entry(val):
    ent = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0}
    for field, _, v in _proto_fields(val):
        if field == 1:
            ent["dtype"] = v
        elif field == 2:
            for f2, _, dim in _proto_fields(v):
                if f2 == 2:  # repeated Dim dim = 2
                    size = 0
                    for f3, _, s in _proto_fields(dim):
                        if f3 == 1:
                            size = s
                    ent["shape"].append(size)
        elif field == 3:
            ent["shard_id"] = v
        elif field == 4:
            ent["offset"] = v
        elif field == 5:
            ent["size"] = v
    return ent


def read_index(index_path):
    """Return {key(str): entry dict} for every tensor in the bundle."""
    buf = open(index_path, "rb").read()
    if struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("not a TensorBundle index (bad magic)")
    footer = buf[-48:]
    pos = 0
    _, pos = _varint(footer, pos)      # metaindex offset
    _, pos = _varint(footer, pos)      # metaindex size
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    out = {}
    for _, handle in _block_entries(buf, idx_off, idx_size):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for key, val in _block_entries(buf, off, size):
            if key == b"":            # bundle header entry
                continue
            out[key.decode()] = _parse_entry(val)
    return out


def read_tensor(prefix, key, index=None):
    """Load one tensor (numpy array) from ``prefix``.index / .data-00000-of-00001."""
    index = index if index is not None else read_index(prefix + ".index")
    ent = index[key]
    if ent["dtype"] not in _DTYPES:
        raise ValueError("dtype enum %d not supported for %s" % (ent["dtype"], key))
    if ent["shard_id"] != 0:
        raise ValueError("multi-shard bundles not supported")
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        f.seek(ent["offset"])
        raw = f.read(ent["size"])
    arr = np.frombuffer(raw, dtype=np.dtype(_DTYPES[ent["dtype"]]).newbyteorder("<"))
    return arr.reshape(ent["shape"]).copy()
