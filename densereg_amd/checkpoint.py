"""TensorFlow checkpoint import / export for the engine.

The reference saves and restores with ``tf.train.Saver(tf.global_variables())`` (model/train_single_gpu.py:108,122,172;
model/test_model.py:33-34), i.e. TF-1.3 "V2" tensor-bundle checkpoints: ``<prefix>.index`` (an SSTable keyed by
variable name, values = ``BundleEntryProto``) + ``<prefix>.data-00000-of-00001`` (raw little-endian tensors).
TensorFlow is not installed here (and must not be needed at inference time), so this module reads and writes the
format directly; it is a restatement of the PUBLISHED format of the pinned dependency ``tensorflow == 1.3``
(tensor_bundle.proto, table/format.cc, table/block.cc of that release):

* SSTable: data blocks + index block + 48-byte footer (two BlockHandles as varint64 pairs, zero padding, magic
  ``0xdb4775248b80fb57``); every block is followed by a 1-byte compression type (0 = none, 1 = snappy) and the
  masked CRC-32C of block + type; block entries are ``shared | unshared | value_len`` varint32 triples with prefix
  compressed keys, then the uint32 restart array and its length.
* key ``""`` -> ``BundleHeaderProto{num_shards=1, endianness=2, version=3}``; key ``<variable name>`` ->
  ``BundleEntryProto{dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32, masked), slices=7}``.
* masked crc = ``rotr(crc, 15) + 0xa282ead8``.

[parity unpinned: no checkpoint produced by the reference is available in this environment; the writer and the
reader are tested against each other, against hand-assembled bytes of the format, and against the CRC-32C and
varint known answers.]

Variable names are the reference's own (``hg_imgproc/Conv/weights``, ``Conv_3/BatchReNorm/moving_mean`` ...), which
are also the engine's parameter names, so mapping is the identity; optimizer slots (``<var>/Adam``, ``<var>/Adam_1``,
``beta1_power``, ``beta2_power``), ``global_step`` and the zero-debias slots of the moving statistics
(``.../moving_mean/biased``, ``.../local_step`` -- TF names them with the variable scope repeated) are recognised.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


class CheckpointError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------
# CRC-32C
# ---------------------------------------------------------------------------------------------
_TABLE: Optional[List[int]] = None
_LIB = None


def _crc_py(data: bytes, crc: int = 0) -> int:
    global _TABLE
    if _TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _TABLE = t
    c = crc ^ 0xFFFFFFFF
    t = _TABLE
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C of a bytes-like object; multi-kilobyte inputs go through the library's ``dr_crc32c`` when it loads."""
    global _LIB
    mv = memoryview(data).cast('B')
    if len(mv) < 4096:
        return _crc_py(mv.tobytes(), crc)
    if _LIB is None:
        try:
            from . import _lib
            _LIB = _lib.load()
        except Exception:                       # the importer also works where the HIP library does not load
            _LIB = False
    if _LIB:
        buf = (C.c_char * len(mv)).from_buffer_copy(mv)
        return int(_LIB.dr_crc32c(crc, buf, len(mv)))
    return _crc_py(mv.tobytes(), crc)


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------
# varints and the two protos
# ---------------------------------------------------------------------------------------------
def put_varint(n: int) -> bytes:
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = val = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError('truncated varint')
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise CheckpointError('varint too long')


def _fields(buf: bytes) -> Iterable[Tuple[int, int, object]]:
    """(field number, wire type, value) of a serialized proto message."""
    pos = 0
    while pos < len(buf):
        key, pos = get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            n, pos = get_varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise CheckpointError('unsupported wire type %d' % wt)
        yield fn, wt, v


def _parse_shape(buf: bytes) -> List[int]:
    dims = []
    for fn, _, v in _fields(buf):
        if fn == 2:                                           # repeated Dim dim = 2 { int64 size = 1; string name = 2 }
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
        elif fn == 3 and v:
            raise CheckpointError('tensor of unknown rank')
    return dims


def parse_entry(buf: bytes) -> dict:
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for fn, _, v in _fields(buf):
        if fn == 1: e['dtype'] = v
        elif fn == 2: e['shape'] = _parse_shape(v)
        elif fn == 3: e['shard_id'] = v
        elif fn == 4: e['offset'] = v
        elif fn == 5: e['size'] = v
        elif fn == 6: e['crc32c'] = struct.unpack('<I', v)[0]
        elif fn == 7: e['sliced'] = True
    return e


def build_entry(dtype_id: int, shape, offset: int, size: int, crc_masked: int) -> bytes:
    shp = b''.join(b'\x12' + put_varint(len(d)) + d for d in (b'\x08' + put_varint(int(s)) for s in shape))
    out = b'\x08' + put_varint(dtype_id) + b'\x12' + put_varint(len(shp)) + shp
    if offset:
        out += b'\x20' + put_varint(offset)
    out += b'\x28' + put_varint(size) + b'\x35' + struct.pack('<I', crc_masked)
    return out


def build_header(num_shards: int = 1, producer: int = 1) -> bytes:
    ver = b'\x08' + put_varint(producer)                      # VersionDef{producer}
    return b'\x08' + put_varint(num_shards) + b'\x1a' + put_varint(len(ver)) + ver     # endianness 0 = little (default)


# ---------------------------------------------------------------------------------------------
# snappy (only decompression: TF writes the bundle index uncompressed, other writers may not)
# ---------------------------------------------------------------------------------------------
def snappy_decompress(buf: bytes) -> bytes:
    n, pos = get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little'); pos += nb
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], 'little'); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little'); pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError('corrupt snappy stream')
        for _ in range(ln):                                   # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError('snappy length mismatch')
    return bytes(out)


# ---------------------------------------------------------------------------------------------
# SSTable
# ---------------------------------------------------------------------------------------------
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
    raw = data[offset:offset + size + 5]
    if len(raw) != size + 5:
        raise CheckpointError('block beyond end of file')
    ctype = raw[size]
    if verify:
        want = unmask_crc(struct.unpack('<I', raw[size + 1:size + 5])[0])
        if crc32c(raw[:size + 1]) != want:
            raise CheckpointError('block checksum mismatch at offset %d' % offset)
    body = raw[:size]
    if ctype == 1:
        body = snappy_decompress(body)
    elif ctype != 0:
        raise CheckpointError('unknown block compression %d' % ctype)
    return body


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise CheckpointError('block too small')
    nrestart = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * nrestart
    if limit < 0:
        raise CheckpointError('bad restart array')
    out, pos, key = [], 0, b''
    while pos < limit:
        shared, pos = get_varint(block, pos)
        unshared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        if shared > len(key):
            raise CheckpointError('bad key prefix')
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    data = open(path, 'rb').read()
    if len(data) < 48:
        raise CheckpointError('%s: not an SSTable (too short)' % path)
    footer = data[-48:]
    if struct.unpack('<Q', footer[40:])[0] != _MAGIC:
        raise CheckpointError('%s: bad table magic' % path)
    _, p = get_varint(footer, 0)                                # metaindex handle (unused)
    _, p = get_varint(footer, p)
    ioff, p = get_varint(footer, p)
    isize, p = get_varint(footer, p)
    entries = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, q = get_varint(handle, 0)
        bsize, q = get_varint(handle, q)
        entries += _block_entries(_read_block(data, boff, bsize, verify))
    return entries


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(prev), len(k))
            while shared < m and prev[shared] == k[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(k) - shared) + put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path: str, items: List[Tuple[bytes, bytes]], block_size: int = 262144):
    """Sorted (key, value) pairs -> an uncompressed SSTable (what TF's BundleWriter produces)."""
    keys = [k for k, _ in items]
    if keys != sorted(keys) or len(set(keys)) != len(keys):
        raise CheckpointError('table keys must be unique and sorted')
    f = bytearray()
    index = []

    def emit(block: bytes) -> Tuple[int, int]:
        off = len(f)
        f.extend(block)
        f.append(0)
        f.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return off, len(block)

    cur, cur_bytes = [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 6
        if cur_bytes >= block_size:
            off, size = emit(_build_block(cur))
            index.append((cur[-1][0], put_varint(off) + put_varint(size)))
            cur, cur_bytes = [], 0
    if cur or not index:
        off, size = emit(_build_block(cur))
        index.append((cur[-1][0] if cur else b'', put_varint(off) + put_varint(size)))
    moff, msize = emit(_build_block([]))                        # empty metaindex block
    ioff, isize = emit(_build_block(index, restart_interval=1))
    footer = put_varint(moff) + put_varint(msize) + put_varint(ioff) + put_varint(isize)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    f.extend(footer)
    with open(path, 'wb') as fh:
        fh.write(bytes(f))


# ---------------------------------------------------------------------------------------------
# tensor bundle
# ---------------------------------------------------------------------------------------------
def _shard_name(prefix: str, shard: int, num: int) -> str:
    return '%s.data-%05d-of-%05d' % (prefix, shard, num)


def read_checkpoint(prefix: str, verify: bool = True, names: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
    """All (or the named) tensors of a V2 checkpoint ``prefix`` (``prefix.index`` must exist)."""
    index = prefix + '.index'
    if not os.path.exists(index):
        raise CheckpointError('%s not found (V1 checkpoints are not supported)' % index)
    entries = read_table(index, verify)
    if not entries or entries[0][0] != b'':
        raise CheckpointError('%s: missing bundle header' % index)
    num_shards, endian = 1, 0
    for fn, _, v in _fields(entries[0][1]):
        if fn == 1: num_shards = v
        elif fn == 2: endian = v
    if endian != 0:
        raise CheckpointError('big-endian bundles are not supported')
    want = set(names) if names is not None else None
    shards: Dict[int, np.memmap] = {}
    out = {}
    for key, val in entries[1:]:
        name = key.decode('utf-8')
        if want is not None and name not in want:
            continue
        e = parse_entry(val)
        if e['sliced']:
            raise CheckpointError('%s: partitioned variables are not supported' % name)
        if e['dtype'] not in _DTYPES:
            raise CheckpointError('%s: unsupported dtype %d' % (name, e['dtype']))
        dt = np.dtype(_DTYPES[e['dtype']])
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise CheckpointError('%s: %d bytes for shape %s of %s' % (name, e['size'], e['shape'], dt))
        sid = e['shard_id']
        if sid not in shards:
            path = _shard_name(prefix, sid, num_shards)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode='r') if os.path.getsize(path) else np.zeros(0, np.uint8)
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size']:
            raise CheckpointError('%s: data shard truncated' % name)
        if verify and e['crc32c'] is not None and crc32c(raw) != unmask_crc(e['crc32c']):
            raise CheckpointError('%s: tensor checksum mismatch' % name)
        out[name] = np.frombuffer(bytes(raw), dtype=dt).reshape(e['shape']).copy()
    if want is not None and want - set(out):
        raise CheckpointError('not in checkpoint: %s' % sorted(want - set(out))[:5])
    return out


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray]):
    """One-shard V2 bundle, tensors in key order (as BundleWriter lays them out)."""
    items = [(b'', build_header())]
    data = bytearray()
    for name in sorted(tensors, key=lambda s: s.encode('utf-8')):
        a = np.asarray(tensors[name])
        if a.dtype not in _DTYPE_IDS:
            raise CheckpointError('%s: dtype %s cannot be written' % (name, a.dtype))
        raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()          # C order, little endian
        items.append((name.encode('utf-8'), build_entry(_DTYPE_IDS[a.dtype], a.shape, len(data), len(raw), mask_crc(crc32c(raw)))))
        data += raw
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(_shard_name(prefix, 0, 1), 'wb') as fh:
        fh.write(bytes(data))
    write_table(prefix + '.index', items)


# ---------------------------------------------------------------------------------------------
# engine <-> checkpoint
# ---------------------------------------------------------------------------------------------
_SLOT_SUFFIXES = ('/biased', '/local_step')


def split_variables(tensors: Dict[str, np.ndarray], param_infos) -> dict:
    """Sort a checkpoint's tensors into what the engine takes.

    ``param_infos`` = ``Handle.param_infos()`` = [(name, shape, trainable)].  Returns a dict with ``model`` (name ->
    array for dr_load_param), ``slots`` (zero-debias slot name as the ENGINE spells it -> array), ``adam_m`` /
    ``adam_v`` (variable name -> array), ``scalars`` (global_step, beta powers), ``unexpected`` and ``missing``.
    """
    shapes = {n: tuple(s) for n, s, _ in param_infos}
    model, slots, adam_m, adam_v, scalars, unexpected = {}, {}, {}, {}, {}, []
    for name, a in tensors.items():
        if name in shapes:
            if tuple(a.shape) != shapes[name] and a.size != int(np.prod(shapes[name])):
                raise CheckpointError('%s: checkpoint shape %s, engine expects %s' % (name, a.shape, shapes[name]))
            model[name] = np.ascontiguousarray(a, np.float32).reshape(shapes[name])
        elif name in ('global_step', 'beta1_power', 'beta2_power'):
            scalars[name] = a
        elif name.endswith('/Adam') and name[:-5] in shapes:
            adam_m[name[:-5]] = np.ascontiguousarray(a, np.float32)
        elif name.endswith('/Adam_1') and name[:-7] in shapes:
            adam_v[name[:-7]] = np.ascontiguousarray(a, np.float32)
        elif name.endswith(_SLOT_SUFFIXES):
            # TF spells "<scope>/<scope>/moving_mean/biased": keep the shortest suffix that names an engine variable
            base, suf = name.rsplit('/', 1)
            parts = base.split('/')
            hit = next(('/'.join(parts[i:]) for i in range(len(parts)) if '/'.join(parts[i:]) in shapes), None)
            if hit is None:
                unexpected.append(name)
            else:
                slots[hit + '/' + suf] = np.ascontiguousarray(a, np.float32).reshape(-1)
        else:
            unexpected.append(name)
    missing = sorted(set(shapes) - set(model))
    return dict(model=model, slots=slots, adam_m=adam_m, adam_v=adam_v, scalars=scalars, unexpected=sorted(unexpected),
                missing=missing)


def load_into(handle, prefix: str, strict: bool = True, verify: bool = True) -> dict:
    """Restore an engine handle (``_lib.Handle``) from a reference checkpoint.  Returns the ``split_variables`` report.
    The caller runs ``dr_finalize_params`` afterwards (``Engine.load_checkpoint`` does)."""
    rep = split_variables(read_checkpoint(prefix, verify), handle.param_infos())
    if strict and rep['missing']:
        raise CheckpointError('checkpoint lacks %d model variables, e.g. %s' % (len(rep['missing']), rep['missing'][:3]))
    for name, _, _ in handle.param_infos():                  # engine order; absent variables keep their initial value
        if name in rep['model']:
            a = rep['model'][name]
            handle.call('dr_load_param', name.encode(), a.ctypes.data, a.size)
    for name, a in rep['slots'].items():                     # after the moving statistics: loading those resets the slots
        try:
            handle.call('dr_load_param', name.encode(), a.ctypes.data, a.size)
        except Exception:
            if strict:
                raise
    return rep


def export_from(handle, prefix: str, global_step: Optional[int] = None, with_slots: bool = True, extra=None):
    """Write the engine's variables as a checkpoint the reference's ``Saver.restore`` reads by name."""
    tensors = dict(handle.read_params())
    if with_slots:
        for name, shape, _ in handle.param_infos():
            if name.endswith(('/moving_mean', '/moving_variance')):
                n = int(np.prod(shape))
                for suf, cnt in (('/biased', n), ('/local_step', 1)):
                    buf = np.empty(cnt, np.float32)
                    try:
                        handle.call('dr_read_param', (name + suf).encode(), buf.ctypes.data, cnt)
                    except Exception:
                        continue                              # inference handle: no slots
                    scope = name.rsplit('/', 1)[0]
                    tensors['%s/%s%s' % (scope, name, suf)] = buf.reshape(shape) if suf == '/biased' else buf.reshape(())
    if global_step is not None:
        # a FLOAT variable in the reference: tf.get_variable('global_step', [], initializer=constant_initializer(0)) has the
        # default dtype float32 (train_single_gpu.py:42), and Saver.restore checks dtypes
        tensors['global_step'] = np.array(global_step, np.float32)
    tensors.update(extra or {})
    write_checkpoint(prefix, tensors)
    return sorted(tensors)
