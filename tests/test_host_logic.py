"""CPU: host-side logic of the package (no GPU): launcher/.ini surface, Darknet file I/O, synthetic
weights, layer vocabulary bookkeeping, result type."""
import os

import numpy as np
import pytest

from helpers import ROOT
from oracle import cases
from tensorflow_yolo_amd import BoundingBox, YoloV2, YoloV3, launcher
from tensorflow_yolo_amd.net import base, layers as PL, synth, v2, v3

CFG_DIR = os.path.join(ROOT, "tensorflow-yolo_amd", "config")


def test_ini_surface_and_path_resolution():
    cfg = launcher.read_config(os.path.join(CFG_DIR, "yolo_3.ini"))
    assert set(cfg) >= {"COMMON", "TEST"}
    t = cfg["TEST"]
    for key in ("image_dir", "out_dir", "batch_size", "threshold", "iou_threshold", "anchors", "class_names",
                "checkpoint_path", "pretrained_weights_path", "cpu_only"):
        assert key in t, key                                   # the keys reference net/yolo.py:42-54 reads
    assert os.path.isabs(t["image_dir"]) and os.path.isabs(t["pretrained_weights_path"])
    assert t["image_dir"].startswith(CFG_DIR)                  # relative to the .ini, not the cwd
    assert isinstance(t["anchors"], list) and len(t["anchors"]) == 18 and len(t["class_names"]) == 80
    assert cfg["COMMON"]["version"] == "v3" and cfg["COMMON"]["input_h"] == "416"
    merged = dict(t)
    merged.update(cfg["COMMON"])
    assert merged["version"] == "v3" and merged["threshold"] == "0.5"
    for name, ver, n in (("yolo_2.ini", "v2", 10), ("yolov2_tiny_voc.ini", "v2-tiny", 10)):
        c = launcher.read_config(os.path.join(CFG_DIR, name))
        assert c["COMMON"]["version"] == ver and len(c["TEST"]["anchors"]) == n


def test_launcher_flags_defaults_and_unsupported_modes(tmp_path):
    with pytest.raises(SystemExit, match="not supported"):
        launcher.main([])                                      # default mode is "anchor", as in the reference
    with pytest.raises(SystemExit, match="not supported"):
        launcher.main(["--mode", "TRAIN"])
    bad = tmp_path / "x.ini"
    bad.write_text("[COMMON]\nversion = v9\n[TEST]\n")
    with pytest.raises(ValueError, match="Unsupported version"):
        launcher.main(["--config", str(bad), "--mode", "test"])
    with pytest.raises(ValueError, match="Unsupported mode"):
        launcher.main(["--config", os.path.join(CFG_DIR, "yolo_3.ini"), "--mode", "bogus"])
    assert isinstance(launcher.pick_model("v2"), YoloV2) and isinstance(launcher.pick_model("v3"), YoloV3)


def test_test_mode_with_no_images_returns_quietly(tmp_path, capsys):
    ini = tmp_path / "t.ini"
    (tmp_path / "imgs").mkdir()
    ini.write_text("[COMMON]\nversion = v3\ninput_h = 96\ninput_w = 96\ninput_c = 3\n[TEST]\nimage_dir = imgs\nout_dir = out\n"
                   "batch_size = 1\nthreshold = 0.5\niou_threshold = 0.6\nanchors = [1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18]\n"
                   "class_names = [\"a\"]\ncheckpoint_path = x\npretrained_weights_path = w\ncpu_only = True\n")
    launcher.main(["--config", str(ini), "--mode", "test"])
    assert "No test images found" in capsys.readouterr().out   # reference net/yolo.py:58-60


def test_darknet_file_roundtrip(tmp_path):
    body = np.arange(37, dtype=np.float32)
    for ver, hdr_bytes in (("v2", 16), ("v3", 20)):            # reference net/v2.py:69-75, net/v3.py:102
        p = str(tmp_path / ("w_%s.weights" % ver))
        base.write_darknet_weights(p, body, ver)
        assert os.path.getsize(p) == hdr_bytes + 4 * 37
        hdr, got = base.read_darknet_weights(p, ver)
        assert np.array_equal(got, body) and len(hdr) == (4 if ver == "v2" else 5)


def test_weight_count_guard_is_stricter_than_reference():
    net = v3.create_network(np.reshape(cases.COCO_V3_ANCHORS, [-1, 2]), ["a"] * 80, False, input_shape=(96, 96, 3))
    with pytest.raises(ValueError, match="weight file holds"):
        v3.attach_weights(net, np.zeros(10, np.float32))
    w = synth.darknet_stream(net, seed=1, num_classes=80)
    assert w.size == 62001757 and w.dtype == np.float32
    assert v3.attach_weights(net, w) == [] and net.darknet_weights is not None
    assert np.array_equal(w, synth.darknet_stream(net, seed=1, num_classes=80))      # seeded


def test_variable_names_follow_darknet_stream_order():
    net = v2.create_full_network(np.reshape(cases.COCO_V2_ANCHORS, [-1, 2]), ["a"] * 80, False)
    convs = [l for l in net if isinstance(l, PL.conv2d_bn_act)]
    assert len(convs) == 23
    assert convs[0].variable_names == ["yolo/conv2d_bn_act_0/%s" % t for t in ("beta", "gamma", "moving_mean", "moving_variance", "kernel")]
    assert convs[-1].variable_names == ["yolo/conv2d_bn_act_22/bias", "yolo/conv2d_bn_act_22/kernel"]
    assert net[0].out.get_shape().as_list() == [None, 416, 416, 3] and net[-1].out.hwc == (13, 13, 425)
    assert net[-1].out.shape[1:] == (13, 13, 425)
    with pytest.raises(NotImplementedError):
        PL.conv2d_bn_act(net[0].out, 8, 3, is_training=True)


def test_yolo_layer_anchor_scaling_and_detection_rows():
    net = v3.create_network(np.reshape(cases.COCO_V3_ANCHORS, [-1, 2]), ["a"] * 80, False, input_shape=(608, 608, 3))
    det = net[-1]
    assert [(y.h, y.w, y.b) for y in det.yolos] == [(19, 19, 3), (38, 38, 3), (76, 76, 3)]
    assert det.yolos[0].anchors[0] == (116 / 32, 90 / 32) and det.yolos[2].anchors[0] == (10 / 8, 13 / 8)
    assert det.out.hwc[0] == 22743


def test_bounding_box_and_unsupported_entries():
    b = BoundingBox(x=0.5, y=0.4, w=0.2, h=0.1, class_idx=3, prob=0.9)
    assert b.get_top_left(100, 200) == ((0.5 - 0.1) * 200, (0.4 - 0.05) * 100)
    assert b.get_bottom_right() == (0.6, 0.45)
    assert base.non_maximum_suppression([], 0.5) == []
    for fn in (YoloV2().train, YoloV3().generate_anchors):
        with pytest.raises(NotImplementedError):
            fn({})


def test_image_preprocess_and_draw(tmp_path):
    from PIL import Image
    p = str(tmp_path / "img.png")
    Image.fromarray((np.random.RandomState(0).rand(30, 40, 3) * 255).astype(np.uint8)).save(p)
    (tmp_path / "skip.txt").write_text("x")
    assert base.load_image_paths(str(tmp_path)) == [p]
    x = base.preprocess_image(p, (16, 24, 3))
    assert x.shape == (16, 24, 3) and x.dtype == np.float64 and 0.0 <= x.min() and x.max() <= 1.0
    batches = list(base.generate_test_batch([p, p, p], 2, (16, 24, 3)))
    assert [b[0].shape[0] for b in batches] == [2, 1]          # last batch short
    out = base.draw_boxes(p, [BoundingBox(x=0.5, y=0.5, w=0.5, h=0.5, class_idx=7, prob=0.8)], ["n%d" % i for i in range(8)])
    base.save_image(out, str(tmp_path / "o" / "img_out.png"))
    assert os.path.exists(str(tmp_path / "o" / "img_out.png"))


def test_tf_checkpoint_bundle_roundtrip_and_darknet_order(tmp_path):
    """net/tfckpt.py: a checkpoint with the reference's variable names (net/layers.py:53-63; kernels HWIO)
    -> the Darknet stream; PARITY UNPINNED against TensorFlow itself (absent): format restated, round trip only"""
    from tensorflow_yolo_amd.net import tfckpt
    names = ["c%d" % i for i in range(20)]
    net = v2.create_tiny_network(np.reshape(cases.VOC_TINY_ANCHORS, [-1, 2]), names, False, input_shape=(96, 96, 3))
    w = synth.darknet_stream(net, seed=3, num_classes=20)
    prefix = str(tmp_path / "yolo-20")
    tfckpt.darknet_to_checkpoint(net, w, prefix)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    b = tfckpt.Bundle(prefix)
    assert "yolo/conv2d_bn_act_0/kernel" in b.entries and "yolo/conv2d_bn_act_8/bias" in b.entries and b.num_shards == 1
    k0 = b.read("yolo/conv2d_bn_act_0/kernel")
    assert k0.shape == (3, 3, 3, 16) and k0.dtype == np.float32                     # HWIO, as tf.layers.conv2d keeps it
    # darknet order: beta, gamma, mean, var, then kernel [out][in][kh][kw] (net/base.py:36-40 transposes it to HWIO)
    assert np.array_equal(np.transpose(k0, (3, 2, 0, 1)).ravel(), w[64:64 + 16 * 27])
    assert np.array_equal(b.read("yolo/conv2d_bn_act_0/gamma"), w[16:32])
    back = tfckpt.checkpoint_to_darknet(net, prefix)
    assert back.dtype == np.float32 and np.array_equal(back, w)
    # a checkpoint of another graph fails like the reference's restore does
    other = v2.create_tiny_network(np.reshape(cases.VOC_TINY_ANCHORS, [-1, 2]), names[:5], False, input_shape=(96, 96, 3))
    with pytest.raises(ValueError, match="shape"):
        tfckpt.checkpoint_to_darknet(other, prefix)
    with pytest.raises(IOError):
        tfckpt.Bundle(str(tmp_path / "missing"))
    # corrupt index block -> checksum error
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[10] ^= 0xff
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tfckpt.Bundle(prefix)


def test_tfckpt_primitives():
    from tensorflow_yolo_amd.net import tfckpt
    assert tfckpt.crc32c(b"123456789") == 0xe3069283                                # the CRC-32C check value
    assert tfckpt._snappy_decompress(b"\x0b\x14hello \x05\x06") == b"hello hello"   # literal + 1-byte-offset copy
    assert tfckpt._snappy_decompress(b"\x03\x08abc") == b"abc"
    for n in (0, 1, 127, 128, 300, 2 ** 40):
        assert tfckpt._varint(tfckpt._put_varint(n), 0) == (n, len(tfckpt._put_varint(n)))


def _crc32c_bitwise(data):
    """CRC-32C restated independently of net/tfckpt.py (bit by bit, reflected polynomial 0x82F63B78)."""
    c = 0xffffffff
    for b in bytearray(data):
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82f63b78 & -(c & 1))
    return c ^ 0xffffffff


def _masked(crc):              # LevelDB/TensorFlow store crcs rotated right by 15 plus a constant (crc32c.h: Mask)
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _vi(n):                    # protobuf / LevelDB varint
    out = b""
    while n >= 0x80:
        out += bytes([(n & 0x7f) | 0x80])
        n >>= 7
    return out + bytes([n])


def test_tf_checkpoint_hand_assembled_index(tmp_path):
    """SURVEY 8f rank 4 / VERDICT r2 #8: a tensor-bundle index assembled BYTE BY BYTE here from the published format
    (LevelDB table: prefix-compressed entries + restart array + 1-byte compression type + masked CRC32C per block, 48-byte
    footer ending in the magic 0xdb4775248b80fb57; tensor_bundle.proto: BundleHeaderProto under key "", BundleEntryProto
    {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32)}) -- nothing from tfckpt.write_bundle.
    It has what a real Saver checkpoint has and the round-trip test lacks: several data blocks, one of them SNAPPY-compressed,
    shared-prefix keys across restart points, keys the graph does not own (global_step as an int64 scalar, Adam slots), two
    data shards.  Still 'parity unpinned' against TensorFlow itself (absent here)."""
    import struct
    from tensorflow_yolo_amd.net import tfckpt
    assert _crc32c_bitwise(b"123456789") == 0xe3069283 == tfckpt.crc32c(b"123456789")

    # a two-conv graph in the reference's naming (net/layers.py:53-63): conv 0 with BN, conv 1 with bias
    PL.conv2d_bn_act.reset()
    net = [PL.input_layer([None, 8, 8, 3])]
    net.append(PL.conv2d_bn_act(net[-1].out, 4, 3, 1, True, "leaky", False))
    net.append(PL.conv2d_bn_act(net[-1].out, 6, 1, 1, False, "linear", False))
    rng = np.random.RandomState(4)
    t = {"yolo/conv2d_bn_act_0/beta": rng.randn(4), "yolo/conv2d_bn_act_0/gamma": rng.rand(4) + 0.5,
         "yolo/conv2d_bn_act_0/moving_mean": rng.randn(4), "yolo/conv2d_bn_act_0/moving_variance": rng.rand(4) + 0.5,
         "yolo/conv2d_bn_act_0/kernel": rng.randn(3, 3, 3, 4), "yolo/conv2d_bn_act_1/bias": rng.randn(6),
         "yolo/conv2d_bn_act_1/kernel": rng.randn(1, 1, 4, 6)}
    t = {k: v.astype("<f4") for k, v in t.items()}
    extra = {"beta1_power": np.array(0.9, "<f4"), "global_step": np.array(1234567, "<i8"),
             "yolo/conv2d_bn_act_0/kernel/Adam": np.zeros((3, 3, 3, 4), "<f4"), "yolo/conv2d_bn_act_0/kernel/Adam_1": np.ones((3, 3, 3, 4), "<f4")}
    everything = dict(t, **extra)
    names = sorted(everything)                      # a table's keys are sorted bytewise
    # two data shards: variables of conv 1 live in shard 1
    shard_of = {n: (1 if "act_1" in n else 0) for n in names}
    blobs, offs = {0: b"", 1: b""}, {}
    for n in names:
        raw = everything[n].tobytes()
        if shard_of[n] == 0 and len(blobs[0]) == 0:
            blobs[0] += b"\0" * 0                   # (first tensor at offset 0: the offset field is then absent, proto3 default)
        offs[n] = len(blobs[shard_of[n]])
        blobs[shard_of[n]] += raw
    prefix = str(tmp_path / "model.ckpt-7")
    for sid in (0, 1):
        open("%s.data-%05d-of-00002" % (prefix, sid), "wb").write(blobs[sid])

    def entry(n):
        a = everything[n]
        dtype_id = {"<f4": 1, "<i8": 9}[a.dtype.str]
        dims = b"".join(b"\x12" + _vi(len(d)) + d for d in (b"\x08" + _vi(s) for s in a.shape))      # TensorShapeProto.dim[].size
        msg = b"\x08" + _vi(dtype_id) + b"\x12" + _vi(len(dims)) + dims
        if shard_of[n]:
            msg += b"\x18" + _vi(shard_of[n])
        if offs[n]:
            msg += b"\x20" + _vi(offs[n])
        raw = a.tobytes()
        return msg + b"\x28" + _vi(len(raw)) + b"\x35" + struct.pack("<I", _masked(_crc32c_bitwise(raw)))

    header = b"\x08\x02" + b"\x1a\x02\x08\x01"       # BundleHeaderProto {num_shards = 2 (field 1), version {producer: 1} (field 3)}; little endian = default
    pairs = [(b"", header)] + [(n.encode(), entry(n)) for n in names]

    def block(items, restart_interval):
        out, restarts, prev = b"", [], b""
        for i, (k, v) in enumerate(items):
            shared = 0
            if i % restart_interval == 0:
                restarts.append(len(out))
            else:
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
            prev = k
        return out + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))

    def snappy(raw):
        """raw Snappy: preamble varint(len), then a literal, one 2-byte-offset copy of a repeated run, and a literal."""
        probe = b"yolo/conv2d_bn_act_0/"
        first = raw.find(probe)
        second = raw.find(probe, first + 1)
        assert first >= 0 and second > first
        ln = len(probe)

        def literal(b):
            assert 0 < len(b) <= 65536
            if len(b) <= 60:
                return bytes([(len(b) - 1) << 2]) + b
            if len(b) <= 256:
                return bytes([60 << 2, len(b) - 1]) + b
            return bytes([61 << 2]) + struct.pack("<H", len(b) - 1) + b
        copy = bytes([((ln - 1) << 2) | 2]) + struct.pack("<H", second - first)
        return _vi(len(raw)) + literal(raw[:second]) + copy + literal(raw[second + ln:])

    def with_trailer(body, ctype):
        return body + bytes([ctype]) + struct.pack("<I", _masked(_crc32c_bitwise(body + bytes([ctype]))))

    # three data blocks: [header + 3 keys] plain, [next 5 keys] SNAPPY with restart interval 1 (full keys repeat the prefix), [rest] plain
    chunks = [(pairs[:4], 16, 0), (pairs[4:9], 1, 1), (pairs[9:], 2, 0)]
    assert all(c[0] for c in chunks) and sum(len(c[0]) for c in chunks) == len(pairs)
    f = b""
    handles = []
    for items, ri, ctype in chunks:
        raw = block(items, ri)
        body = snappy(raw) if ctype else raw
        handles.append((items[-1][0], _vi(len(f)) + _vi(len(body))))
        f += with_trailer(body, ctype)
    meta = block([], 16)
    meta_handle = _vi(len(f)) + _vi(len(meta))
    f += with_trailer(meta, 0)
    index = block(handles, 1)
    footer = meta_handle + _vi(len(f)) + _vi(len(index))
    f += with_trailer(index, 0)
    f += footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    open(prefix + ".index", "wb").write(f)

    b = tfckpt.Bundle(prefix)
    assert b.num_shards == 2 and sorted(b.entries) == names
    for n in names:
        got = b.read(n)
        assert got.dtype == everything[n].dtype.newbyteorder("=") and got.shape == everything[n].shape and np.array_equal(got, everything[n]), n
    assert int(b.read("global_step")) == 1234567
    # the Darknet stream in layer-list order: beta, gamma, mean, var, kernel [out][in][kh][kw] | bias, kernel (net/base.py:26-46)
    want = np.concatenate([t["yolo/conv2d_bn_act_0/" + k] for k in ("beta", "gamma", "moving_mean", "moving_variance")] +
                          [np.transpose(t["yolo/conv2d_bn_act_0/kernel"], (3, 2, 0, 1)).ravel(), t["yolo/conv2d_bn_act_1/bias"],
                           np.transpose(t["yolo/conv2d_bn_act_1/kernel"], (3, 2, 0, 1)).ravel()])
    assert np.array_equal(tfckpt.checkpoint_to_darknet(net, prefix), want)
    # a flipped byte of tensor DATA is caught by the entry's crc32c (corrupted shards must not load silently)
    raw = bytearray(blobs[1]); raw[5] ^= 0x40
    open("%s.data-00001-of-00002" % prefix, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="data checksum"):
        tfckpt.Bundle(prefix).read("yolo/conv2d_bn_act_1/bias")
    # ... and a flipped byte inside the COMPRESSED block by the block trailer
    bad = bytearray(f); bad[len(with_trailer(block(chunks[0][0], 16), 0)) + 7] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="checksum"):
        tfckpt.Bundle(prefix)


def test_draw_boxes_content(tmp_path):
    """SURVEY 8f rank 2 (reference net/base.py:212-226): the rectangle of thickness 3 sits on the box edges at the coordinates
    scaled by the ORIGINAL image size, top-left clamped at 0, in COLORS[class_idx % 6] (the reference's BGR tuples, so RGB-swapped
    here), the label text starts at the top-left corner, and every pixel away from boxes and labels is untouched."""
    from PIL import Image
    p = str(tmp_path / "flat.png")
    W, H = 200, 120
    Image.new("RGB", (W, H), (17, 17, 17)).save(p)
    names = ["n%d" % i for i in range(9)]
    boxes = [BoundingBox(x=0.5, y=0.5, w=0.4, h=0.5, class_idx=7, prob=0.8),          # inside: tl (60, 30), br (140, 90)
             BoundingBox(x=0.1, y=0.2, w=0.3, h=0.6, class_idx=2, prob=0.5)]          # tl (-10, -12) -> clamped (0, 0), br (50, 60)
    img = np.asarray(base.draw_boxes(p, boxes, names))
    assert img.shape == (H, W, 3)

    def rgb(ci):
        b, g, r = base.COLORS[ci % len(base.COLORS)]
        return (r, g, b)
    c7, c2 = rgb(7), rgb(2)
    assert c7 != c2 and len(base.COLORS) == 6
    # box 1: cv2.rectangle thickness 3 = the corner coordinate +- 1: rows 29..31 and 89..91, columns 59..61 and 139..141
    for y in (29, 30, 31, 89, 90, 91):
        assert all(tuple(img[y, x]) == c7 for x in range(59, 142)), y
    for x in (59, 60, 61, 139, 140, 141):
        assert all(tuple(img[y, x]) == c7 for y in range(29, 92)), x
    assert tuple(img[28, 100]) == (17, 17, 17) and tuple(img[32, 100]) == (17, 17, 17) and tuple(img[60, 100]) == (17, 17, 17)
    assert tuple(img[60, 58]) == (17, 17, 17) and tuple(img[60, 62]) == (17, 17, 17) and tuple(img[60, 142]) == (17, 17, 17)
    # box 2: top-left clamped at the image corner (the part at -1 falls off the image), bottom-right at (50, 60)
    for x in range(0, 52):
        assert all(tuple(img[y, x]) == c2 for y in (59, 60, 61)), x
    for y in range(2, 62):
        assert all(tuple(img[y, x]) == c2 for x in (49, 50, 51)), y
        assert tuple(img[y, 0]) == c2 and tuple(img[y, 1]) == c2, y
    assert tuple(img[30, 25]) == (17, 17, 17)
    # labels: baseline at (tl.x, tl.y - 10): box 1's text sits in rows ~9..20 right of x = 60; box 2's baseline is at y = -10,
    # off the image -- the reference loses that label too
    lab1 = img[6:22, 60:140].reshape(-1, 3).astype(int)
    changed = (lab1 != 17).any(axis=1)
    assert changed.sum() >= 40                                                          # "n7 0.800" (anti-aliased glyphs)
    toward = (lab1[changed] - 17) * (np.array(c7) - 17)                                 # every touched pixel moves towards the class colour
    assert (toward >= 0).all() and (toward > 0).any(axis=1).all()
    assert (img[23:28, 62:139] == 17).all()                                             # nothing between the baseline and the box
    # everything outside the two outlines' bounding rectangles and the label strip is the original image
    mask = np.ones((H, W), bool)
    mask[29:92, 59:142] = False; mask[0:62, 0:52] = False; mask[4:24, 59:142] = False
    assert (img[mask] == 17).all()
    inner = np.zeros((H, W), bool)
    inner[32:89, 62:139] = True; inner[2:59, 2:49] = True                              # box interiors stay as they were
    assert (img[inner] == 17).all()


def test_tap_variant_tables_agree_with_the_instantiation_list():
    """conv_tap.hip: kTapNB / kTapTP (what conv_tap_fits sizes the position-interleaved patch planes with) against the template arguments of
    YOLO_TAP_VARIANTS -- positions per tile = WN x TP x 16; a wrong TP lets a map through whose patch does not fit its planes (round 5: the
    256 x 224 tile is 2 x 7 fragments wide, not 224 / 64)."""
    import os
    import re
    src = open(os.path.join(ROOT, "tensorflow-yolo_amd", "csrc", "conv_tap.hip")).read()
    tp = [int(v) for v in re.search(r"kTapTP\[\] = \{([^}]*)\}", src).group(1).split(",")]
    nb = [int(v) for v in re.search(r"kTapNB\[\] = \{([^}]*)\}", src).group(1).split(",")]
    prg = [int(v) for v in re.search(r"kTapPRG\[\] = \{([^}]*)\}", src).group(1).split(",")]
    block = src[src.index("#define YOLO_TAP_VARIANTS(X)"):src.index("const char *conv_tap_symbol")]
    seen = 0
    for m in re.finditer(r"X\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)", block):
        i, wm, wn, tm, t, g, occ, mode = (int(v) for v in m.groups())
        assert tp[i] == t and nb[i] == wn * t * 16 and prg[i] == g, (i, tp[i], t, nb[i], wn * t * 16, prg[i], g)
        seen += 1
    assert seen == len(tp) == len(nb) == len(prg) == 14
