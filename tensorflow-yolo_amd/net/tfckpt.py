"""TensorFlow checkpoint (tensor bundle, "V2" format) reader -- without TensorFlow.

The reference restores `checkpoint_path` with `tf.train.Saver().restore` before it falls back to the Darknet weights
(reference net/yolo.py:71-78, net/base.py:55-61).  A checkpoint written by its TRAIN mode (`saver.save`, net/yolo.py:173)
is a tensor bundle:

    <prefix>.index                   an SSTable (LevelDB table format): key = variable name -> BundleEntryProto
                                     {dtype, shape, shard_id, offset, size, crc32c}; key "" -> BundleHeaderProto
    <prefix>.data-0000i-of-0000n     the raw little-endian tensor bytes

and the variables of a YOLO graph are named `yolo/conv2d_bn_act_<i>/{kernel,bias,beta,gamma,moving_mean,moving_variance}`
(reference net/layers.py:53-63 `variable_names`; kernels are HWIO).  `checkpoint_to_darknet` turns such a bundle into the
Darknet float stream the rest of the package consumes (kernel back to [out][in][kh][kw], the inverse of net/base.py:36-40),
so a restored checkpoint and a .weights file take the same path to the GPU.

PARITY UNPINNED: TensorFlow is not installable here and the reference ships no checkpoint, so the format is restated from
TensorFlow's published sources (tensor_bundle.proto, core/lib/io/format.cc, table_builder.cc) and checked by round trip
against `write_bundle` below (same restatement) only.  Index blocks may be Snappy-compressed (decoder included); block CRCs of the index
are always verified; the CRC of the tensor DATA is verified for tensors up to `VERIFY_DATA_CRC_BYTES` (bias / BN vectors, small
kernels) and for every tensor with `Bundle(prefix, verify_data=True)` (pure-Python CRC32C over hundreds of MB is slow).
tests/test_host_logic.py::test_tf_checkpoint_hand_assembled_index parses an index assembled byte by byte in the test from the
published format (its own CRC and Snappy encoder), independent of `write_bundle`.
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 19: np.float16}     # tensorflow DataType enum
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}
VERIFY_DATA_CRC_BYTES = 64 << 10        # tensors up to this size always have their data CRC checked


# ---- little helpers: varints, protobuf wire format, crc32c ------------------------------------------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _put_varint(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _proto_fields(buf):
    """[(field number, wire type, value)] of one serialized message (varint / 64-bit / bytes / 32-bit)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v, pos = bytes(buf[pos:pos + n]), pos + n
        elif wt == 5:
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), as LevelDB/TensorFlow use it (table-driven, pure Python: small inputs only)."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    c = crc ^ 0xffffffff
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _snappy_decompress(buf):
    """Raw Snappy block format (the index blocks of some writers)."""
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy stream")
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ---- SSTable (core/lib/io/table*.cc, the LevelDB table format) -----------------------------------------------------------------
def _read_block(f, offset, size):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block")
    body, ctype, crc = raw[:size], raw[size], struct.unpack_from("<I", raw, size + 1)[0]
    if _mask(crc32c(raw[:size + 1])) != crc:
        raise ValueError("table block checksum mismatch")
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return body


def _block_entries(block):
    """(key, value) pairs of one block: prefix-compressed keys, restart array at the end."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def _read_table(path):
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 48:
            raise ValueError("%s is too short to be a table" % path)
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != _MAGIC:
            raise ValueError("%s: not a TensorFlow checkpoint index (bad magic)" % path)
        _, pos = _varint(footer, 0)             # metaindex handle (offset, size): unused
        _, pos = _varint(footer, pos)
        ioff, pos = _varint(footer, pos)
        isize, pos = _varint(footer, pos)
        entries = []
        for _, handle in _block_entries(_read_block(f, ioff, isize)):
            boff, p2 = _varint(handle, 0)
            bsize, _ = _varint(handle, p2)
            entries.extend(_block_entries(_read_block(f, boff, bsize)))
        return entries


# ---- tensor bundle ---------------------------------------------------------------------------------------------------------------
def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for num, wt, v in _proto_fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:                          # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for n2, _, d in _proto_fields(v):
                if n2 == 2:
                    e["shape"].append(next((x for k, _, x in _proto_fields(d) if k == 1), 0))
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["sliced"] = True
    return e


class Bundle(object):
    """Index of one checkpoint; tensors are read on demand."""

    def __init__(self, prefix, verify_data=False):
        self.prefix = prefix
        self.verify_data = bool(verify_data)
        index = prefix + ".index"
        if not os.path.exists(index):
            raise IOError("%s not found" % index)
        self.entries, self.num_shards = {}, 1
        for key, value in _read_table(index):
            if key == b"":
                hdr = dict((n, v) for n, _, v in _proto_fields(value))
                self.num_shards = hdr.get(1, 1)
                if hdr.get(2, 0) != 0:
                    raise ValueError("big-endian checkpoints are not supported")
            else:
                self.entries[key.decode()] = _parse_entry(value)

    def names(self):
        return sorted(self.entries)

    def read(self, name):
        e = self.entries[name]
        if e["sliced"]:
            raise ValueError("%s is stored in slices (partitioned variable): not supported" % name)
        if e["dtype"] not in _DTYPES:
            raise ValueError("%s: unsupported dtype %d" % (name, e["dtype"]))
        dt = np.dtype(_DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise ValueError("%s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
        path = "%s.data-%05d-of-%05d" % (self.prefix, e["shard_id"], self.num_shards)
        with open(path, "rb") as f:
            f.seek(e["offset"])
            arr = np.fromfile(f, dtype=dt.newbyteorder("<"), count=count)
        if arr.size != count:
            raise ValueError("%s: data shard %s is truncated" % (name, path))
        if e["crc32c"] is not None and (self.verify_data or e["size"] <= VERIFY_DATA_CRC_BYTES):
            if _mask(crc32c(arr.tobytes())) != e["crc32c"]:
                raise ValueError("%s: tensor data checksum mismatch in %s" % (name, path))
        return arr.reshape(e["shape"])


def checkpoint_to_darknet(net, prefix):
    """The float32 Darknet stream (layer-list order: per conv beta, gamma, moving_mean, moving_variance | bias, then the
    kernel as [out][in][kh][kw]) from the variables a restore of `prefix` would assign (reference net/layers.py:53-63).
    Raises when the checkpoint lacks a variable or a shape differs -- the reference's restore fails there too."""
    bundle = Bundle(prefix)
    parts = []
    for layer in net:
        for name in layer.variable_names:
            if name not in bundle.entries:
                raise KeyError("checkpoint has no variable %s" % name)
            arr = np.asarray(bundle.read(name), dtype=np.float32)
            if name.rsplit("/", 1)[1] == "kernel":
                want = (layer.ksize, layer.ksize, layer.in_channels, layer.filters)        # HWIO, as tf.layers.conv2d stores it
                if tuple(arr.shape) != want:
                    raise ValueError("%s has shape %s, the graph needs %s" % (name, tuple(arr.shape), want))
                arr = np.transpose(arr, (3, 2, 0, 1))                                       # inverse of net/base.py:40
            elif arr.shape != (layer.filters,):
                raise ValueError("%s has shape %s, the graph needs (%d,)" % (name, tuple(arr.shape), layer.filters))
            parts.append(np.ascontiguousarray(arr).ravel())
    return np.concatenate(parts) if parts else np.zeros(0, np.float32)


# ---- writer (tests, and converting Darknet weights into a checkpoint TensorFlow's format describes) --------------------------------
def _entry_proto(dtype_id, shape, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype_id) + b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


def _build_block(pairs, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors):
    """tensors: {variable name: ndarray}.  One data shard, uncompressed index blocks, CRCs filled in."""
    names = sorted(tensors)
    data_path = "%s.data-00000-of-00001" % prefix
    pairs = [(b"", b"\x08\x01\x1a\x02\x08\x01")]         # BundleHeaderProto {num_shards: 1, version {producer: 1}}
    offset = 0
    with open(data_path, "wb") as f:
        for n in names:
            a = np.ascontiguousarray(tensors[n])
            if a.dtype not in _DTYPE_IDS:
                raise ValueError("unsupported dtype %s" % a.dtype)
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            pairs.append((n.encode(), _entry_proto(_DTYPE_IDS[a.dtype], a.shape, offset, len(raw), _mask(crc32c(raw)))))
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        handles, pos = [], 0
        for start in range(0, len(pairs), 64):          # data blocks of 64 entries
            chunk = pairs[start:start + 64]
            block = _build_block(chunk)
            trailer = b"\x00" + struct.pack("<I", _mask(crc32c(block + b"\x00")))
            f.write(block + trailer)
            handles.append((chunk[-1][0], _put_varint(pos) + _put_varint(len(block))))
            pos += len(block) + 5
        meta = _build_block([])
        f.write(meta + b"\x00" + struct.pack("<I", _mask(crc32c(meta + b"\x00"))))
        meta_handle = _put_varint(pos) + _put_varint(len(meta))
        pos += len(meta) + 5
        index = _build_block(handles, restart_interval=1)
        f.write(index + b"\x00" + struct.pack("<I", _mask(crc32c(index + b"\x00"))))
        footer = meta_handle + _put_varint(pos) + _put_varint(len(index))
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))


def darknet_to_checkpoint(net, weights, prefix):
    """The inverse of checkpoint_to_darknet: the variables a TRAIN-mode `saver.save` of this graph would hold."""
    weights = np.asarray(weights, dtype=np.float32)
    tensors, pos = {}, 0
    for layer in net:
        for name in layer.variable_names:
            if name.rsplit("/", 1)[1] == "kernel":
                n = layer.filters * layer.in_channels * layer.ksize * layer.ksize
                oihw = weights[pos:pos + n].reshape(layer.filters, layer.in_channels, layer.ksize, layer.ksize)
                tensors[name] = np.ascontiguousarray(np.transpose(oihw, (2, 3, 1, 0)))     # net/base.py:40
            else:
                n = layer.filters
                tensors[name] = weights[pos:pos + n].copy()
            pos += n
    if pos != weights.size:
        raise ValueError("weight stream holds %d values, the graph needs %d" % (weights.size, pos))
    write_bundle(prefix, tensors)
