"""TFRecord files and ``tf.train.Example`` messages without TensorFlow.

The reference stores every dataset as TFRecord shards of serialized ``tf.train.Example`` protos
(``data/dataset_base.py:49-62`` writer, ``:166-178`` reader; features ``name`` (bytes), ``xyz_pose`` (floats),
``png16`` (bytes) and, for the NYU test set, ``bbx`` (floats): ``data/nyu.py:158-177``).

* record framing [TF format, ``tensorflow/core/lib/io/record_writer.cc``]: ``uint64 length | uint32 masked_crc32c(length)
  | data | uint32 masked_crc32c(data)``, little endian, the mask being ``rot15(crc) + 0xa282ead8`` -- the same CRC-32C the
  checkpoint files use (``densereg_amd/checkpoint.py``; large payloads go through ``dr_crc32c``).
* ``Example{1: Features{1: map<string, Feature>}}``, ``Feature{1: BytesList | 2: FloatList | 3: Int64List}``, each list
  ``{1: repeated value}`` -- floats packed (wire type 2) as TF writes them, unpacked (wire type 5) accepted.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, Iterator, List, Union

import numpy as np

from ..checkpoint import _fields, crc32c, get_varint, mask_crc, put_varint


class RecordError(ValueError):
    pass


# ---- framing ---------------------------------------------------------------------------------
def read_records(path: str, verify: bool = True) -> Iterator[bytes]:
    """Yield the payload of every record of a TFRecord file (``tf.TFRecordReader``)."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise RecordError('%s: truncated record header' % path)
            n, = struct.unpack('<Q', head[:8])
            if verify and mask_crc(crc32c(head[:8])) != struct.unpack('<I', head[8:])[0]:
                raise RecordError('%s: corrupted record length' % path)
            body = f.read(n + 4)
            if len(body) != n + 4:
                raise RecordError('%s: truncated record' % path)
            if verify and mask_crc(crc32c(body[:n])) != struct.unpack('<I', body[n:])[0]:
                raise RecordError('%s: corrupted record data' % path)
            yield body[:n]


def write_records(path: str, records: Iterable[bytes]) -> int:
    """``tf.python_io.TFRecordWriter``: returns the number of records written."""
    k = 0
    with open(path, 'wb') as f:
        for r in records:
            r = bytes(r)
            ln = struct.pack('<Q', len(r))
            f.write(ln + struct.pack('<I', mask_crc(crc32c(ln))) + r + struct.pack('<I', mask_crc(crc32c(r))))
            k += 1
    return k


# ---- tf.train.Example --------------------------------------------------------------------------
Value = Union[List[bytes], np.ndarray]


def _ld(field: int, payload: bytes) -> bytes:
    return put_varint((field << 3) | 2) + put_varint(len(payload)) + payload


def make_example(features: Dict[str, Value]) -> bytes:
    """Serialize {name: bytes | [bytes] | float array | int array}; keys are written in sorted order (proto map
    order is unspecified; sorted is what the python protobuf runtime of TF 1.3 produces with deterministic output)."""
    out = b''
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b''.join(_ld(1, bytes(b)) for b in v))                       # BytesList
        else:
            a = np.asarray(v)
            if a.dtype.kind == 'f':
                feat = _ld(2, _ld(1, np.ascontiguousarray(a, '<f4').tobytes()))        # FloatList, packed
            elif a.dtype.kind in 'iu':
                feat = _ld(3, _ld(1, b''.join(put_varint(int(x) & 0xFFFFFFFFFFFFFFFF) for x in a.ravel())))
            else:
                raise RecordError('feature %r: unsupported value type %s' % (key, a.dtype))
        entry = _ld(1, key.encode()) + _ld(2, feat)
        out += _ld(1, entry)                                                           # Features.feature map entry
    return _ld(1, out)                                                                 # Example.features


def parse_example(buf: bytes) -> Dict[str, Value]:
    """``tf.parse_single_example`` without a schema: bytes features -> list of bytes, float -> float32 array,
    int64 -> int64 array."""
    out: Dict[str, Value] = {}
    for fn, wt, features in _fields(buf):
        if fn != 1 or wt != 2:
            continue
        for fn2, wt2, entry in _fields(features):
            if fn2 != 1 or wt2 != 2:
                continue
            key, feat = None, b''
            for fn3, wt3, v in _fields(entry):
                if fn3 == 1:
                    key = bytes(v).decode()
                elif fn3 == 2:
                    feat = v
            if key is None:
                raise RecordError('feature map entry without a key')
            val: Value = []
            for kind, wtk, lst in _fields(feat):
                if kind == 1:
                    val = [bytes(v) for f, _, v in _fields(lst) if f == 1]
                elif kind == 2:
                    parts = []
                    for f, w, v in _fields(lst):
                        if f == 1:
                            parts.append(np.frombuffer(bytes(v), '<f4'))                # packed run or one fixed32
                    val = np.concatenate(parts) if parts else np.zeros(0, np.float32)
                elif kind == 3:
                    ints: List[int] = []
                    for f, w, v in _fields(lst):
                        if f != 1:
                            continue
                        if w == 0:
                            ints.append(v)
                        else:
                            pos = 0
                            while pos < len(v):
                                x, pos = get_varint(v, pos)
                                ints.append(x)
                    val = np.array([x - (1 << 64) if x >= 1 << 63 else x for x in ints], np.int64)
            out[key] = val
    return out
