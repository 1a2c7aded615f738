"""TensorFlow checkpoint (tensor-bundle V2) reader / writer without TensorFlow -- SURVEY §8f-2.

The reference saves `tf.train.Saver` checkpoints (`pointnet2/train.py:197-199`, `model.ckpt.index` +
`model.ckpt.data-00000-of-00001`) and publishes pre-trained ones (`README.md:116-117`).  Variable names here ARE the
reference's (`graph.py` keeps the TF scope names: `layer1/conv0/weights`, `layer1/conv0/bn/moving_mean`, `fc1/biases`, ...)
and variables keep their TF shapes ([1,1,Cin,Cout] conv kernels), so importing is a name-for-name copy.

File format, restated from the TensorFlow sources it is defined by (tensorflow/core/util/tensor_bundle/tensor_bundle.cc,
tensorflow/core/lib/io/{table_builder,format,block_builder}.cc -- the LevelDB table format):
  <prefix>.index                 an SSTable: data blocks | metaindex block | index block | 48-byte footer
      block   = entries, uint32 restart offsets, uint32 restart count; then a 5-byte trailer: compression type (0 = none)
                + masked CRC-32C of (block ‖ type)
      entry   = varint32 shared-key-bytes, varint32 unshared-key-bytes, varint32 value-bytes, key suffix, value
      footer  = metaindex BlockHandle, index BlockHandle (varint64 offset, size), zero padding to 40 bytes,
                magic 0xdb4775248b80fb57 (little endian)
      key ""  -> BundleHeaderProto {1: num_shards, 2: endianness, 3: VersionDef}
      key v   -> BundleEntryProto  {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: fixed32 masked crc32c}
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset+size)
PARITY NOTE: no TensorFlow and no TensorFlow-written checkpoint exist in this environment.  The writer is pinned by a
hand-assembled byte image; the reader additionally by a bundle assembled in the test from the format description by an
independent code path (bit-serial CRC, own varint / protobuf / block code, two data shards, multi-block index with
prefix compression, reversed and unknown proto fields) -- tests/test_tf_checkpoint_cpu.py.  Not against a real file.
"""
import re
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
_DTYPE_ID = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---------------------------------------------------------------------------------------------- CRC-32C (Castagnoli)
def _make_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _make_table()


def crc32c(data):
    crc = 0xFFFFFFFF
    for b in bytes(data):
        crc = _TAB[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- varints / protobuf
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _parse_proto(buf):
    """{field: [values]}: varints as int, fixed32 as int, length-delimited as bytes"""
    out, pos = {}, 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _get_varint(buf, pos)
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        out.setdefault(field, []).append(v)
    return out


def _field(field, wire, payload):
    return _put_varint((field << 3) | wire) + payload


def _entry_proto(dtype_id, shape, offset, size, crc):
    dims = b"".join(_field(2, 2, _put_varint(len(d)) + d) for d in (_field(1, 0, _put_varint(int(s))) for s in shape))
    msg = _field(1, 0, _put_varint(dtype_id)) + _field(2, 2, _put_varint(len(dims)) + dims)
    # shard_id 0 is the proto default and omitted, like every zero-valued scalar field
    if offset:
        msg += _field(4, 0, _put_varint(offset))
    msg += _field(5, 0, _put_varint(size)) + _field(6, 5, struct.pack("<I", crc))
    return msg


# ---------------------------------------------------------------------------------------------- SSTable
def _read_block(buf, offset, size, verify):
    body, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(trailer) != 5:
        raise ValueError("truncated table block")
    if trailer[0] != 0:
        raise ValueError("compressed table block (type %d): not supported" % trailer[0])
    if verify and struct.unpack("<I", trailer[1:])[0] != masked_crc32c(body + trailer[:1]):
        raise ValueError("table block checksum mismatch at offset %d" % offset)
    nrestart = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * nrestart
    entries, pos, key = [], 0, b""
    while pos < end:
        shared, pos = _get_varint(body, pos)
        unshared, pos = _get_varint(body, pos)
        vlen, pos = _get_varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + unshared])
        pos += unshared
        entries.append((key, bytes(body[pos:pos + vlen])))
        pos += vlen
    return entries


def _handle(buf, pos):
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


def read_index(path, verify=True):
    """-> (header fields, {name: (dtype, shape, shard, offset, size, masked crc)})"""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = buf[-48:]
    _mo, _ms, pos = _handle(footer, 0)
    io, isz, _ = _handle(footer, pos)
    header, entries = None, {}
    for _sep, hv in _read_block(buf, io, isz, verify):
        bo, bs, _ = _handle(hv, 0)
        for key, val in _read_block(buf, bo, bs, verify):
            f = _parse_proto(val)
            if key == b"":
                header = f
                continue
            shape = tuple(_parse_proto(d).get(1, [0])[0] for d in _parse_proto(f.get(2, [b""])[0]).get(2, []))
            entries[key.decode()] = (f.get(1, [0])[0], shape, f.get(3, [0])[0], f.get(4, [0])[0], f.get(5, [0])[0],
                                     f.get(6, [0])[0])
            if 7 in f:
                raise ValueError("variable %s is stored in slices (partitioned variable): not supported" % key.decode())
    if header is None:
        raise ValueError("checkpoint index has no header entry")
    if header.get(2, [0])[0] != 0:
        raise ValueError("big-endian checkpoint: not supported")
    return header, entries


def read_checkpoint(prefix, verify=False):
    """{variable name: numpy array}.  verify=True also checks every tensor's CRC-32C (slow in pure Python)."""
    header, entries = read_index(prefix + ".index", verify=True)
    nshards = header.get(1, [1])[0]
    shards = {}
    out = {}
    for name, (dt, shape, shard, offset, size, crc) in entries.items():
        if dt not in _DTYPES:
            raise ValueError("variable %s has unsupported dtype id %d" % (name, dt))
        if shard not in shards:
            shards[shard] = np.fromfile("%s.data-%05d-of-%05d" % (prefix, shard, nshards), dtype=np.uint8)
        raw = shards[shard][offset:offset + size]
        if raw.size != size:
            raise ValueError("variable %s: data shard too short" % name)
        if verify and masked_crc32c(raw.tobytes()) != crc:
            raise ValueError("variable %s: tensor checksum mismatch" % name)
        arr = raw.view(_DTYPES[dt])
        if int(np.prod(shape, dtype=np.int64)) != arr.size:
            raise ValueError("variable %s: %d elements stored, shape %s" % (name, arr.size, shape))
        out[name] = arr.reshape(shape).copy()
    return out


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_checkpoint(prefix, variables, block_size=4096):
    """variables: {name: array}; writes <prefix>.index and <prefix>.data-00000-of-00001 in key order"""
    items = sorted(((k.encode(), np.asarray(v)) for k, v in variables.items()), key=lambda kv: kv[0])
    data, table_entries = bytearray(), []
    header = _field(1, 0, _put_varint(1)) + _field(3, 2, _put_varint(2) + _field(1, 0, _put_varint(1)))
    table_entries.append((b"", header))
    for key, arr in items:
        if arr.dtype not in _DTYPE_ID:
            raise ValueError("%s: dtype %s has no TensorFlow id here" % (key.decode(), arr.dtype))
        raw = arr.tobytes()
        table_entries.append((key, _entry_proto(_DTYPE_ID[arr.dtype], arr.shape, len(data), len(raw), masked_crc32c(raw))))
        data += raw
    out, index = bytearray(), _BlockBuilder(1)

    def emit(block):
        off = len(out)
        out.extend(block + b"\x00" + struct.pack("<I", masked_crc32c(block + b"\x00")))
        return _put_varint(off) + _put_varint(len(block))

    bb = _BlockBuilder(16)
    for key, val in table_entries:
        bb.add(key, val)
        if len(bb.buf) >= block_size:
            index.add(bb.last, emit(bb.finish()))
            bb = _BlockBuilder(16)
    if bb.count:
        index.add(bb.last, emit(bb.finish()))
    meta = emit(_BlockBuilder(16).finish())
    idx = emit(index.finish())
    footer = meta + idx
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))
    open(prefix + ".index", "wb").write(bytes(out))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))


# ---------------------------------------------------------------------------------------------- name map
# DGCNN's non-distributed BN keeps its moving statistics in ExponentialMovingAverage shadow variables whose names repeat
# the scope (dgcnn/utils/tf_util.py:484-506); everything else is name-for-name.
_ALIASES = [
    (re.compile(r"^(.*)/bn/\1/bn/moments/Squeeze/ExponentialMovingAverage$"), r"\1/bn/moving_mean"),
    (re.compile(r"^(.*)/bn/\1/bn/moments/Squeeze_1/ExponentialMovingAverage$"), r"\1/bn/moving_variance"),
]
_SKIP = re.compile(r"(^|/)(Adam(_\d+)?|beta[12]_power|global_step|Variable(_\d+)?)$")


def tf_name_to_key(name, prefix="graph."):
    """state_dict key for a TF variable name, or None for optimizer slots / step counters"""
    if _SKIP.search(name):
        return None
    for pat, rep in _ALIASES:
        if pat.match(name):
            name = pat.sub(rep, name)
            break
    return prefix + name


def load_tf_checkpoint(net, ckpt_prefix, strict=True, verify=False):
    """copy a TF checkpoint into `net` (a graph.Model); returns (loaded keys, missing keys, unexpected TF names)"""
    import torch
    sd = net.state_dict()
    loaded, unexpected = [], []
    for name, arr in read_checkpoint(ckpt_prefix, verify=verify).items():
        key = tf_name_to_key(name)
        if key is None:
            continue
        if key not in sd:
            unexpected.append(name)
            continue
        if tuple(sd[key].shape) != tuple(arr.shape):
            raise ValueError("%s: checkpoint shape %s, model shape %s" % (name, arr.shape, tuple(sd[key].shape)))
        with torch.no_grad():
            sd[key].copy_(torch.from_numpy(arr).to(sd[key].dtype))
        loaded.append(key)
    missing = [k for k in sd if k not in set(loaded)]
    if strict and (missing or unexpected):
        raise KeyError("checkpoint / model mismatch: missing %s, unexpected %s" % (missing[:8], unexpected[:8]))
    return loaded, missing, unexpected


def save_tf_checkpoint(net, ckpt_prefix, prefix="graph."):
    """write `net`'s variables under their TF names (restorable by the reference's tf.train.Saver)"""
    write_checkpoint(ckpt_prefix, {k[len(prefix):]: v.detach().cpu().numpy() for k, v in net.state_dict().items()
                                   if k.startswith(prefix)})
