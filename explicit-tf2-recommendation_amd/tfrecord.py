"""Input side of the hot path: the reference's on-disk format, without TensorFlow.

    2.FM/DataGenerator.py:76-90    get_feature_dims / encode_and_record: per-column LabelEncoder (sorted classes -> 0..n-1)
                                   plus the column's offset = sum of the earlier columns' class counts
    2.FM/DataGenerator.py:104-124  write_tf_records: one tf.train.Example per row -- "label" float_list [1] and one
                                   int64_list [1] per encoded column -- through tf.io.TFRecordWriter (2.FM/Tools.py:8-54)
    2.FM/DataGenerator.py:126-134  data_info.json = [dims, offsets, total]
    2.FM/ModelManager.py:122-153   init_dataset: TFRecordDataset -> parse_single_example(FixedLenFeature([1])) -> batch

Everything here is integer / byte work on the host (the reference does it with pandas, sklearn and tf.data); the batches
it yields are the dicts of [B,1] arrays every layer and engine of this package takes.  File format (public TensorFlow
spec): record = uint64 length | masked crc32c(length) | data | masked crc32c(data), little endian,
mask(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32; data = a serialized tf.train.Example protobuf:
Example{1: Features{1: map<string, Feature{1: BytesList | 2: FloatList | 3: Int64List}>}}, numeric lists packed.
No data files ship with the reference, so this row is pinned by the published known answers of CRC-32C, by hand-assembled
protobuf bytes and by write -> read round trips (tests/test_tfrecord.py): "parity unpinned" against TF-written files.
"""
import os
import struct

import numpy as np

# ---------------------------------------------------------------------------------------------------
# CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), table driven
# ---------------------------------------------------------------------------------------------------
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)
_TABLE = tuple(_TABLE)


def crc32c(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    tab = _TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------
# protobuf wire format: just what tf.train.Example needs
# ---------------------------------------------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1                                      # int64 -> two's complement, 10 bytes when negative
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _ld(field, payload):                                    # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """features: {name: float | int | list of floats | list of ints | bytes}.  Python ints -> Int64List, floats ->
    FloatList, bytes -> BytesList (what DataGenerator.write_tf_records builds, 2.FM/DataGenerator.py:111-116)."""
    entries = b""
    for name in features:                                   # insertion order (any order parses)
        v = features[name]
        vals = list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v]
        if isinstance(vals[0], (bytes, bytearray)):
            feat = _ld(1, b"".join(_ld(1, bytes(x)) for x in vals))
        elif isinstance(vals[0], (float, np.floating)):
            feat = _ld(2, _ld(1, struct.pack("<%df" % len(vals), *[float(x) for x in vals])))
        else:
            feat = _ld(3, _ld(1, b"".join(_varint(int(x)) for x in vals)))
        entries += _ld(1, _ld(1, name.encode()) + _ld(2, feat))
    return _ld(1, entries)


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield field, wt, val


def decode_example(buf):
    """Serialized tf.train.Example -> {name: np.float32 array | np.int64 array | list of bytes}."""
    out = {}
    for f1, _, features in _fields(buf):
        if f1 != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            name, feat = None, b""
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    name = bytes(v).decode()
                elif f3 == 2:
                    feat = v
            val = None
            for kind, _, lst in _fields(feat):
                if kind == 1:                               # BytesList
                    val = [bytes(v) for f, _, v in _fields(lst) if f == 1]
                elif kind == 2:                             # FloatList: packed (wire type 2) or repeated fixed32
                    acc = []
                    for f, wt, v in _fields(lst):
                        if f == 1:
                            acc.append(np.frombuffer(bytes(v), "<f4"))
                    val = np.concatenate(acc) if acc else np.zeros(0, np.float32)
                elif kind == 3:                             # Int64List: packed varints or repeated varint
                    acc = []
                    for f, wt, v in _fields(lst):
                        if f != 1:
                            continue
                        if wt == 0:
                            acc.append(v)
                        else:
                            p, b = 0, bytes(v)
                            while p < len(b):
                                x, p = _read_varint(b, p)
                                acc.append(x)
                    a = np.array(acc, dtype=np.uint64).astype(np.int64) if acc else np.zeros(0, np.int64)
                    val = a
            if name is not None:
                out[name] = val
    return out


# ---------------------------------------------------------------------------------------------------
# TFRecord files
# ---------------------------------------------------------------------------------------------------
class TFRecordWriter:
    """tf.io.TFRecordWriter(path).write(bytes) (2.FM/Tools.py:33,38,45,51)."""

    def __init__(self, path):
        self.f = open(path, "wb")

    def write(self, data):
        head = struct.pack("<Q", len(data))
        self.f.write(head + struct.pack("<I", masked_crc(head)) + data + struct.pack("<I", masked_crc(data)))

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_records(path, verify=True):
    """Yields the payload of every record of one TFRecord file; a bad CRC or a truncated record raises ValueError
    (tf.data raises DataLossError)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) < 8:
                raise ValueError("%s: truncated record header" % path)
            (n,) = struct.unpack("<Q", head)
            crc_h = f.read(4)
            data = f.read(n)
            crc_d = f.read(4)
            if len(crc_h) < 4 or len(data) < n or len(crc_d) < 4:
                raise ValueError("%s: truncated record" % path)
            if verify and (struct.unpack("<I", crc_h)[0] != masked_crc(head) or
                           struct.unpack("<I", crc_d)[0] != masked_crc(data)):
                raise ValueError("%s: corrupted record (CRC mismatch)" % path)
            yield data


# ---------------------------------------------------------------------------------------------------
# DataGenerator: label encoding with per-column offsets, data_info.json
# ---------------------------------------------------------------------------------------------------
def label_encode_columns(columns):
    """columns: {name: 1-D array of raw values}, in encode_columns order.  Returns (encoded {name: int64 array},
    recorder {name: {str(value): id}}, data_info [dims, offsets, total]) -- sklearn's LabelEncoder sorts the classes
    (np.unique), ids are class rank + the column's offset (2.FM/DataGenerator.py:76-90, 126-134)."""
    names = list(columns)
    classes = {n: np.unique(np.asarray(columns[n])) for n in names}
    dims = [int(len(classes[n])) for n in names]
    offsets = [0] + [int(x) for x in np.cumsum(dims[:-1])]
    encoded, recorder = {}, {}
    for n, off in zip(names, offsets):
        encoded[n] = np.searchsorted(classes[n], np.asarray(columns[n])).astype(np.int64) + off
        recorder[n] = {str(v): int(i + off) for i, v in enumerate(classes[n].tolist())}
    return encoded, recorder, [dims, offsets, int(sum(dims))]


def write_dataset(output_path, data_name, encoded, labels, data_type, feature_names, doc_limit=200000):
    """write_tf_records (2.FM/DataGenerator.py:104-124) with CustomTFWriter's file naming and roll-over
    (2.FM/Tools.py:8-54: '<data_name>-<train|test>-<k>', a new file once the inner counter reaches doc_limit)."""
    os.makedirs(output_path, exist_ok=True)
    state = {}
    for t in ("train", "test"):
        state[t] = [1, 1, TFRecordWriter(os.path.join(output_path, "%s-%s-%d" % (data_name, t, 1)))]
    counter = {"train": 0, "test": 0}
    for i in range(len(labels)):
        t = data_type[i]
        if t not in state:
            continue
        sample = {"label": float(labels[i])}
        for n in feature_names:
            sample[n] = int(encoded[n][i])
        st = state[t]
        st[1] += 1
        st[2].write(encode_example(sample))
        if st[1] >= doc_limit:
            st[2].close()
            st[0] += 1
            st[2] = TFRecordWriter(os.path.join(output_path, "%s-%s-%d" % (data_name, t, st[0])))
            st[1] = 1
        counter[t] += 1
    for t in state:
        state[t][2].close()
    return counter


class TFRecordDataset:
    """init_dataset (2.FM/ModelManager.py:122-153): every file of data_dir whose name contains `mode`, each record
    parsed as FixedLenFeature([1]) (float32 label, int64 features), batched; the last batch may be short.  Iterating
    yields {name: int64 [b,1], label_name: float32 [b,1]} numpy batches (data.to_device moves them to the GPU)."""

    def __init__(self, data_dir, mode, feature_names, label_name="label", batch=100):
        assert mode in ("train", "test")
        self.files = sorted(os.path.join(data_dir, f) for f in os.listdir(data_dir) if mode in f)
        self.feature_names, self.label_name, self.batch = list(feature_names), label_name, int(batch)

    def __iter__(self):
        cols = {n: [] for n in self.feature_names}
        lab = []

        def flush():
            out = {n: np.array(cols[n], np.int64).reshape(-1, 1) for n in self.feature_names}
            out[self.label_name] = np.array(lab, np.float32).reshape(-1, 1)
            for n in cols:
                cols[n].clear()
            lab.clear()
            return out

        for path in self.files:
            for rec in read_records(path):
                ex = decode_example(rec)
                for n in self.feature_names:
                    v = ex.get(n)
                    if v is None or len(v) != 1:
                        raise ValueError("feature %r: expected exactly one int64 (FixedLenFeature([1]))" % n)
                    cols[n].append(int(v[0]))
                v = ex.get(self.label_name)
                if v is None or len(v) != 1:
                    raise ValueError("label %r: expected exactly one float" % self.label_name)
                lab.append(float(v[0]))
                if len(lab) == self.batch:
                    yield flush()
        if lab:
            yield flush()
