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
