"""A TensorFlow tensor bundle (checkpoint V2) assembled from the public format description alone -- deliberately NOT
with scanobjectnn_amd.tf_checkpoint's writer or any of its helpers: bit-serial CRC-32C, its own varint / protobuf /
block code, and choices the in-tree writer never makes (restart interval 3, tiny data blocks -> many blocks and a
multi-entry index block, two data shards, proto fields emitted in reverse order, an unknown extra field, a non-empty
metaindex block).  It stands in for a TensorFlow-written file, which cannot be produced here (TensorFlow is absent and
there is no network): the reader is pinned against THIS code path (tests/test_tf_checkpoint_cpu.py) and the GPU restore
test (tests/test_checkpoint_eval_gpu.py) feeds the product models from it."""
import struct

import numpy as np

def _indep_crc32c(data):
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def _indep_mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _vi(n):
    out = b""
    while True:
        b7, n = n & 0x7F, n >> 7
        out += bytes([b7 | (0x80 if n else 0)])
        if not n:
            return out


def _pb(num, wire, payload):
    return _vi(num << 3 | wire) + payload


def _ld(num, payload):
    return _pb(num, 2, _vi(len(payload)) + payload)


def _indep_block(items, restart_interval):
    body, restarts, prev = b"", [], None
    for n, (key, val) in enumerate(items):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(key), len(prev)) and key[shared] == prev[shared]:
                shared += 1
        body += _vi(shared) + _vi(len(key) - shared) + _vi(len(val)) + key[shared:] + val
        prev = key
    body += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return body + b"\x00" + struct.pack("<I", _indep_mask(_indep_crc32c(body + b"\x00")))


def _indep_bundle(prefix, tensors, nshards=2, entries_per_block=2):
    names = sorted(tensors)
    shard_bytes = [b""] * nshards
    items = []
    version = _pb(1, 0, _vi(1))                                        # VersionDef.producer = 1
    items.append((b"", _ld(3, version) + _pb(2, 0, _vi(0)) + _pb(1, 0, _vi(nshards))))   # header, fields 3, 2, 1
    dtype_id = {"float32": 1, "int32": 3, "int64": 9}
    for i, name in enumerate(names):
        arr = np.asarray(tensors[name])              # (ascontiguousarray would turn the scalar into shape (1,))
        raw = arr.tobytes()
        shard = i % nshards
        offset = len(shard_bytes[shard])
        shard_bytes[shard] += raw + b"\xAB" * (i % 3)                 # unreferenced filler between tensors
        shape = b"".join(_ld(2, _pb(1, 0, _vi(d))) for d in arr.shape)
        # BundleEntryProto with the fields in REVERSE order + an unknown varint field 15 a reader has to skip
        entry = (_pb(15, 0, _vi(7)) + _pb(6, 5, struct.pack("<I", _indep_mask(_indep_crc32c(raw)))) +
                 _pb(5, 0, _vi(len(raw))) + _pb(4, 0, _vi(offset)) + _pb(3, 0, _vi(shard)) + _ld(2, shape) +
                 _pb(1, 0, _vi(dtype_id[str(arr.dtype)])))
        items.append((name.encode(), entry))
    table, index_items = b"", []
    for k in range(0, len(items), entries_per_block):
        chunk = items[k:k + entries_per_block]
        blk = _indep_block(chunk, restart_interval=3)
        index_items.append((chunk[-1][0] + b"\x00", _vi(len(table)) + _vi(len(blk) - 5)))   # separator >= last key
        table += blk
    meta = _indep_block([(b"filter.none", b"")], 1)
    meta_handle = _vi(len(table)) + _vi(len(meta) - 5)
    table += meta
    index = _indep_block(index_items, restart_interval=1)
    index_handle = _vi(len(table)) + _vi(len(index) - 5)
    table += index
    footer = meta_handle + index_handle
    table += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    with open(prefix + ".index", "wb") as f:
        f.write(table)
    for s in range(nshards):
        with open("%s.data-%05d-of-%05d" % (prefix, s, nshards), "wb") as f:
            f.write(shard_bytes[s])
