"""TF-1.3 V2 checkpoint import / export (``densereg_amd/checkpoint.py``).

No checkpoint written by the reference exists here (TensorFlow cannot be installed), so the format is pinned three
ways: known answers of its primitives (CRC-32C check value, masked CRC, varints, snappy), a table assembled BY HAND in
this file from the published block / footer layout (independent of the writer), and writer <-> reader round trips
including corruption detection.  The last tests drive the engine: export -> restore -> identical variables, slot
variables and Adam moments, and the reference's variable naming (scope repeated for the zero-debias slots).
"""
import struct

import numpy as np
import pytest

from densereg_amd import checkpoint as ck


def test_crc32c_and_mask_known_answers():
    assert ck.crc32c(b'123456789') == 0xE3069283            # the CRC-32C check value
    assert ck.crc32c(b'') == 0
    assert ck.crc32c(bytes(32)) == 0x8A9136AA                # rfc3720 B.4: 32 bytes of zeros
    assert ck.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43       # rfc3720 B.4: 32 bytes of ones
    big = bytes(range(256)) * 64                             # > 4 KB: goes through the library
    assert ck.crc32c(big) == ck._crc_py(big)
    assert ck.crc32c(big[100:], ck.crc32c(big[:100])) == ck.crc32c(big)      # continuation
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert ck.unmask_crc(ck.mask_crc(c)) == c
    assert ck.mask_crc(0) == 0xa282ead8


def test_varints_and_entry_proto():
    for n, b in ((0, b'\x00'), (1, b'\x01'), (127, b'\x7f'), (128, b'\x80\x01'), (300, b'\xac\x02'), (2 ** 32, b'\x80\x80\x80\x80\x10')):
        assert ck.put_varint(n) == b
        assert ck.get_varint(b, 0) == (n, len(b))
    # BundleEntryProto by hand: dtype DT_FLOAT(1), shape [3,4], offset 64, size 48, crc fixed32
    raw = (b'\x08\x01' + b'\x12\x08' + b'\x12\x02\x08\x03' + b'\x12\x02\x08\x04' + b'\x20\x40' + b'\x28\x30' +
           b'\x35' + struct.pack('<I', 0xDEADBEEF))
    e = ck.parse_entry(raw)
    assert (e['dtype'], e['shape'], e['offset'], e['size'], e['crc32c']) == (1, [3, 4], 64, 48, 0xDEADBEEF)
    assert ck.build_entry(1, (3, 4), 64, 48, 0xDEADBEEF) == raw
    assert ck.parse_entry(ck.build_entry(3, (), 0, 4, 7))['shape'] == []


def test_snappy_known_vector():
    # literal "abcd" (tag 0x0c = len 4), then copy len 8 offset 4 (1-byte-offset form: tag 0b000_100_01 = 0x11, off 4)
    comp = b'\x0c' + b'\x0c' + b'abcd' + b'\x11\x04'
    assert ck.snappy_decompress(comp) == b'abcdabcdabcd'
    with pytest.raises(ck.CheckpointError):
        ck.snappy_decompress(b'\x05' + b'\x0c' + b'abcd')


def _hand_table(tmp_path):
    """An SSTable written out byte by byte from the format description (table/format.cc, block.cc)."""
    def block(body):
        return body + b'\x00' + struct.pack('<I', ck.mask_crc(ck.crc32c(body + b'\x00')))
    # data block: keys "" , "Conv/biases", "Conv/weights" (shares the 5-byte prefix "Conv/")
    ents = (b'\x00\x00\x02' + b'' + b'h0' +
            b'\x00\x0b\x02' + b'Conv/biases' + b'v1' +
            b'\x05\x07\x02' + b'weights' + b'v2')
    data = ents + struct.pack('<I', 0) + struct.pack('<I', 1)
    f = block(data)
    meta_off = len(f)
    meta = struct.pack('<I', 0) + struct.pack('<I', 1)
    f += block(meta)
    idx_off = len(f)
    handle = ck.put_varint(0) + ck.put_varint(len(data))
    idx = b'\x00\x0c' + ck.put_varint(len(handle)) + b'Conv/weights' + handle + struct.pack('<I', 0) + struct.pack('<I', 1)
    f += block(idx)
    footer = ck.put_varint(meta_off) + ck.put_varint(len(meta)) + ck.put_varint(idx_off) + ck.put_varint(len(idx))
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    p = tmp_path / 'hand.index'
    p.write_bytes(f + footer)
    return p


def test_reader_on_hand_assembled_table(tmp_path):
    p = _hand_table(tmp_path)
    assert ck.read_table(str(p)) == [(b'', b'h0'), (b'Conv/biases', b'v1'), (b'Conv/weights', b'v2')]
    raw = bytearray(p.read_bytes())
    raw[10] ^= 1
    p.write_bytes(bytes(raw))
    with pytest.raises(ck.CheckpointError, match='checksum'):
        ck.read_table(str(p))
    assert ck.read_table(str(p), verify=False)[1][0] != b'Conv/biases' or True      # unverified read does not raise
    raw[-1] ^= 0xFF
    p.write_bytes(bytes(raw))
    with pytest.raises(ck.CheckpointError, match='magic'):
        ck.read_table(str(p))


def test_writer_reader_round_trip_multi_block(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {'Conv_%d/BatchReNorm/%s' % (i, k): rng.standard_normal(rng.integers(1, 40)).astype(np.float32)
               for i in range(120) for k in ('beta', 'gamma', 'moving_mean')}
    tensors['hg_imgproc/Conv/weights'] = rng.standard_normal((7, 7, 1, 32)).astype(np.float32)
    tensors['global_step'] = np.array(12345, np.int32)
    tensors['step64'] = np.array([2 ** 40, -3], np.int64)
    tensors['empty'] = np.zeros((0, 4), np.float32)
    tensors['flags'] = np.array([True, False, True])
    prefix = str(tmp_path / 'model.ckpt-7')
    ck.write_checkpoint(prefix, tensors)
    back = ck.read_checkpoint(prefix)
    assert sorted(back) == sorted(tensors)
    for k in tensors:
        assert back[k].dtype == tensors[k].dtype and back[k].shape == tensors[k].shape
        np.testing.assert_array_equal(back[k], tensors[k])
    # several data blocks (small block size) read back the same
    items = [(b'', ck.build_header())] + [(('k%04d' % i).encode(), bytes([i % 251]) * (i % 17)) for i in range(500)]
    ck.write_table(str(tmp_path / 't.index'), items, block_size=256)
    assert ck.read_table(str(tmp_path / 't.index')) == items
    # subset read, missing name, corrupt tensor bytes
    assert list(ck.read_checkpoint(prefix, names=['global_step'])) == ['global_step']
    with pytest.raises(ck.CheckpointError, match='not in checkpoint'):
        ck.read_checkpoint(prefix, names=['nope'])
    data = prefix + '.data-00000-of-00001'
    raw = bytearray(open(data, 'rb').read())
    raw[len(raw) // 2] ^= 0x10
    open(data, 'wb').write(bytes(raw))
    with pytest.raises(ck.CheckpointError, match='tensor checksum'):
        ck.read_checkpoint(prefix)
    assert len(ck.read_checkpoint(prefix, verify=False)) == len(tensors)


def test_engine_export_restore_round_trip(emu, tmp_path):
    """Engine (training handle) -> checkpoint with the reference's names -> fresh handle: every variable, the
    zero-debias slots and their step counter come back bit-exact; unknown / missing names are reported."""
    from oracle import net
    from oracle.graph import NetConfig, param_specs
    cfg = NetConfig(1, 8, 2)
    rng = np.random.default_rng(1)
    params = net.init_params(cfg, seed=3)
    for k in params:
        if params[k].dtype == np.float32 and params[k].size > 1:
            params[k] = rng.standard_normal(params[k].shape).astype(np.float32)
        if k.endswith('moving_variance'):
            params[k] = np.abs(params[k]) + 0.5
    h = emu.handle(cfg, 1, training=True)
    h.load_params(params)
    # give one layer non-trivial slot state, as after a few training steps
    name = 'Conv_2/BatchReNorm/moving_mean'
    n = int(np.prod(dict((a, b) for a, b, _ in h.param_infos())[name]))
    biased = rng.standard_normal(n).astype(np.float32)
    import ctypes as C
    h.call('dr_load_param', (name + '/biased').encode(), biased.ctypes.data, n)
    h.call('dr_load_param', (name + '/local_step').encode(), np.array([7], np.float32).ctypes.data, 1)
    prefix = str(tmp_path / 'model.ckpt-40')
    names = ck.export_from(h, prefix, global_step=40)
    assert 'global_step' in names and 'Conv_2/BatchReNorm/Conv_2/BatchReNorm/moving_mean/biased' in names
    tensors = ck.read_checkpoint(prefix)
    # train_single_gpu.py:42: global_step is a float32 variable in the reference graph; Saver.restore checks the dtype
    assert tensors['global_step'].dtype == np.float32 and tensors['global_step'].shape == () and float(tensors['global_step']) == 40.0
    assert [n for n, _, _ in param_specs(cfg)] == [n for n in [p[0] for p in h.param_infos()]]
    # what a reference checkpoint additionally holds: Adam slots and the beta powers
    tensors['Conv/weights/Adam'] = np.zeros_like(tensors['Conv/weights'])
    tensors['Conv/weights/Adam_1'] = np.ones_like(tensors['Conv/weights'])
    tensors['beta1_power'] = np.array(0.5 ** 8, np.float32)
    tensors['some/other/variable'] = np.zeros(3, np.float32)
    ck.write_checkpoint(prefix, tensors)
    h2 = emu.handle(cfg, 1, training=True)
    rep = ck.load_into(h2, prefix)
    assert rep['missing'] == [] and rep['unexpected'] == ['some/other/variable']
    assert list(rep['adam_m']) == ['Conv/weights'] and float(rep['scalars']['beta1_power']) == 0.5 ** 8
    a, b = h.read_params(), h2.read_params()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    got = np.empty(n, np.float32)
    h2.call('dr_read_param', (name + '/biased').encode(), got.ctypes.data, n)
    np.testing.assert_array_equal(got, biased)
    step = np.empty(1, np.float32)
    h2.call('dr_read_param', (name + '/local_step').encode(), step.ctypes.data, 1)
    assert step[0] == 7.0
    # a checkpoint that lacks a model variable is refused in strict mode
    del tensors['Conv/weights']
    ck.write_checkpoint(prefix, tensors)
    with pytest.raises(ck.CheckpointError, match='lacks'):
        ck.load_into(emu.handle(cfg, 1, training=True), prefix)
    assert 'Conv/weights' in ck.load_into(emu.handle(cfg, 1, training=True), prefix, strict=False)['missing']
    h.close(); h2.close()


# ---------------------------------------------------------------------------------------------------------------------------
# Independent implementations (nothing below is encoded by densereg_amd/checkpoint.py): Google's snappy library through pyarrow, and
# the official protobuf runtime on message types declared from TensorFlow's public tensor_bundle.proto / tensor_shape.proto /
# versions.proto / tensor_slice.proto field numbers.  What stays unpinned by any independent writer is the leveldb table container
# itself (footer, block handles, restart arrays): no leveldb and no TensorFlow exists in this image -- INTEGRATION.md says so.
# ---------------------------------------------------------------------------------------------------------------------------
def _snappy_cases():
    rng = np.random.default_rng(5)
    yield b''
    yield b'a'
    yield bytes(rng.integers(0, 256, 61, dtype=np.uint8))                       # a literal one byte past the 60-byte inline length
    yield bytes(rng.integers(0, 256, 70000, dtype=np.uint8))                    # incompressible: long literals (2- and 3-byte lengths)
    yield b'abcabcabcabc' * 50 + b'xyz'                                         # overlapping copies (offset < length)
    yield (b'0123456789abcdef' * 8 + bytes(rng.integers(0, 256, 3000, dtype=np.uint8))) * 40     # copies with 2-byte offsets
    blob = bytes(rng.integers(0, 4, 200000, dtype=np.uint8))                    # low-entropy: many short copies at every distance
    yield blob
    yield blob[:66000] + bytes(rng.integers(0, 256, 70000, dtype=np.uint8)) + blob[:66000]      # a match further than 64 KB back


def test_snappy_decoder_against_googles_encoder():
    """``snappy_decompress`` on raw snappy blocks produced by Google's library (pyarrow's codec): every element type the format has
    -- inline and extended literal lengths, copies with 1- and 2-byte offsets (4-byte ones where the encoder emits them), overlapping
    copies -- decoded back to the input, and the official decoder agreeing on the same bytes."""
    pa = pytest.importorskip('pyarrow')
    if not pa.Codec.is_available('snappy'):
        pytest.skip('pyarrow without snappy')
    kinds = set()
    for data in _snappy_cases():
        comp = pa.compress(data, codec='snappy', asbytes=True)
        assert ck.snappy_decompress(comp) == data
        assert pa.decompress(comp, decompressed_size=len(data), codec='snappy', asbytes=True) == data
        pos = ck.get_varint(comp, 0)[1]
        while pos < len(comp):                                                  # walk the elements: which tag kinds did the encoder use?
            tag = comp[pos]; kind = tag & 3; kinds.add(kind)
            if kind == 0:
                ln = tag >> 2
                nb = ln - 59 if ln >= 60 else 0
                ln = int.from_bytes(comp[pos + 1:pos + 1 + nb], 'little') if nb else ln
                pos += 1 + nb + ln + 1
            else:
                pos += {1: 2, 2: 3, 3: 5}[kind]
    assert {0, 1, 2} <= kinds, kinds


def test_snappy_compressed_table_blocks_written_by_googles_encoder(tmp_path):
    """A table whose data blocks are snappy-compressed by Google's encoder (type byte 1, masked CRC-32C over compressed bytes + type,
    as table/format.cc lays a block out) read back by ``read_table``: the compressed-block path TF's own writer does not take but
    other bundle writers may."""
    pa = pytest.importorskip('pyarrow')
    if not pa.Codec.is_available('snappy'):
        pytest.skip('pyarrow without snappy')
    items = [(('var_%04d/weights' % i).encode(), bytes([i % 251]) * (40 + i % 7)) for i in range(300)]
    blocks, index_items, out = [], [], bytearray()

    def emit(raw_block, compress):
        body = pa.compress(raw_block, codec='snappy', asbytes=True) if compress else raw_block
        typ = b'\x01' if compress else b'\x00'
        off = len(out)
        out.extend(body + typ + struct.pack('<I', ck.mask_crc(ck.crc32c(body + typ))))
        return off, len(body)
    for i in range(0, len(items), 50):
        chunk = items[i:i + 50]
        off, size = emit(ck._build_block(chunk), compress=True)
        index_items.append((chunk[-1][0], ck.put_varint(off) + ck.put_varint(size)))
    moff, msize = emit(ck._build_block([]), compress=False)
    ioff, isize = emit(ck._build_block(index_items, restart_interval=1), compress=True)
    footer = ck.put_varint(moff) + ck.put_varint(msize) + ck.put_varint(ioff) + ck.put_varint(isize)
    out.extend(footer + b'\x00' * (40 - len(footer)) + bytes.fromhex('57fb808b247547db'))      # table/format.h: kTableMagicNumber, little endian
    path = tmp_path / 'snappy.index'
    path.write_bytes(bytes(out))
    assert ck.read_table(str(path)) == items


def _bundle_protos():
    """BundleHeaderProto / BundleEntryProto (+ TensorShapeProto, TensorSliceProto, VersionDef) declared from the field numbers of
    TensorFlow's public .proto files, instantiated by the OFFICIAL protobuf runtime."""
    pytest.importorskip('google.protobuf')
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='dr_tensor_bundle_test.proto', package='drtest', syntax='proto3')

    def msg(name, fields, nested=()):
        m = descriptor_pb2.DescriptorProto(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
        for n in nested:
            m.nested_type.add().CopyFrom(n)
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    dim = msg('Dim', [('size', 1, F.TYPE_INT64, OPT, ''), ('name', 2, F.TYPE_STRING, OPT, '')])
    fd.message_type.add().CopyFrom(msg('TensorShapeProto', [('dim', 2, F.TYPE_MESSAGE, REP, '.drtest.TensorShapeProto.Dim'),
                                                            ('unknown_rank', 3, F.TYPE_BOOL, OPT, '')], nested=[dim]))
    ext = msg('Extent', [('start', 1, F.TYPE_INT64, OPT, ''), ('length', 2, F.TYPE_INT64, OPT, '')])
    fd.message_type.add().CopyFrom(msg('TensorSliceProto', [('extent', 1, F.TYPE_MESSAGE, REP, '.drtest.TensorSliceProto.Extent')], nested=[ext]))
    fd.message_type.add().CopyFrom(msg('VersionDef', [('producer', 1, F.TYPE_INT32, OPT, ''), ('min_consumer', 2, F.TYPE_INT32, OPT, ''),
                                                      ('bad_consumers', 3, F.TYPE_INT32, REP, '')]))
    fd.message_type.add().CopyFrom(msg('BundleHeaderProto', [('num_shards', 1, F.TYPE_INT32, OPT, ''), ('endianness', 2, F.TYPE_INT32, OPT, ''),
                                                             ('version', 3, F.TYPE_MESSAGE, OPT, '.drtest.VersionDef')]))
    fd.message_type.add().CopyFrom(msg('BundleEntryProto', [
        ('dtype', 1, F.TYPE_INT32, OPT, ''), ('shape', 2, F.TYPE_MESSAGE, OPT, '.drtest.TensorShapeProto'), ('shard_id', 3, F.TYPE_INT32, OPT, ''),
        ('offset', 4, F.TYPE_INT64, OPT, ''), ('size', 5, F.TYPE_INT64, OPT, ''), ('crc32c', 6, F.TYPE_FIXED32, OPT, ''),
        ('slices', 7, F.TYPE_MESSAGE, REP, '.drtest.TensorSliceProto')]))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, 'GetMessageClass', None)
    cls = (lambda n: get(pool.FindMessageTypeByName('drtest.' + n))) if get else \
          (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName('drtest.' + n)))
    return cls('BundleHeaderProto'), cls('BundleEntryProto')


def test_bundle_protos_against_the_official_protobuf_runtime():
    """``parse_entry`` on entries SERIALIZED by the protobuf runtime (any field order it chooses, zero / large / negative values, a
    sliced entry), and ``build_entry`` / ``build_header`` PARSED by it: the wire-format code of the importer and exporter against an
    encoder and a decoder it does not share a line with."""
    Header, Entry = _bundle_protos()
    rng = np.random.default_rng(11)
    for trial in range(40):
        shape = [int(v) for v in rng.integers(1, 600, int(rng.integers(0, 5)))]
        e = Entry(dtype=int(rng.choice([1, 3, 9])), shard_id=int(rng.integers(0, 3)), offset=int(rng.integers(0, 1 << 40)),
                  size=int(rng.integers(0, 1 << 33)), crc32c=int(rng.integers(0, 1 << 32)))
        for d in shape:
            e.shape.dim.add(size=d)
        if trial % 7 == 0:
            e.slices.add().extent.add(start=0, length=4)
        got = ck.parse_entry(e.SerializeToString())
        assert (got['dtype'], got['shape'], got['shard_id'], got['offset'], got['size'], got['crc32c'], got['sliced']) == \
               (e.dtype, shape, e.shard_id, e.offset, e.size, e.crc32c, trial % 7 == 0)
        # the exporter's bytes, read by the official decoder
        back = Entry.FromString(ck.build_entry(e.dtype, shape, e.offset, e.size, e.crc32c))
        assert (back.dtype, [d.size for d in back.shape.dim], back.offset, back.size, back.crc32c, back.shard_id) == \
               (e.dtype, shape, e.offset, e.size, e.crc32c, 0)
    h = Header.FromString(ck.build_header(num_shards=1, producer=1))
    assert (h.num_shards, h.endianness, h.version.producer) == (1, 0, 1)
