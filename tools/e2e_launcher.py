#!/usr/bin/env python3
"""End-to-end TEST mode: `launcher --mode test` on a generated directory of image files --
Pillow decode -> device resize -> forward + decode + NMS -> records D2H -> draw -> encode + write -- serial loop against the
three-stage pipeline of Yolo._test_pipelined, same files and console lines.

    python tools/e2e_launcher.py [--images 256] [--out gpurun_out/e2e.json]

The images are the committed fixture tests/golden/dog_416_rgb_u8.npz (the reference's img/dog.jpg decoded and stretched in the build
container; /root/reference does not exist on the GPU box) tiled up to 768 x 576 -- the size of the original file -- and written as
JPEG and as PNG under different names.  Networks: the reference's config/yolo_2.ini [TEST_COCO] values (YOLOv2 416, threshold 0.5,
IoU 0.6) at fp32 / batch 1 as shipped there and at fp16 / batch 16, and config/yolo_3.ini's (YOLOv3 608) at fp16 / batch 32, on
synthetic Darknet weights.  Prints one JSON line per run and writes them to --out."""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_dir(root, n, ext):
    from PIL import Image
    px = np.load(os.path.join(ROOT, "tests", "golden", "dog_416_rgb_u8.npz"))["rgb"]
    big = np.tile(px, (2, 2, 1))[:576, :768]
    d = os.path.join(root, "img_" + ext)
    os.makedirs(d)
    rng = np.random.RandomState(3)
    for i in range(n):
        im = big.copy()
        im[:8, :8] = rng.randint(0, 256, size=(8, 8, 3))         # (files differ)
        Image.fromarray(im).save(os.path.join(d, "dog%04d.%s" % (i, ext)), quality=90) if ext == "jpg" else \
            Image.fromarray(im).save(os.path.join(d, "dog%04d.%s" % (i, ext)))
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    from tensorflow_yolo_amd import launcher
    from tensorflow_yolo_amd.net import base, synth
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        dirs = {ext: make_dir(tmp, args.images, ext) for ext in ("jpg", "png")}
        for kind, size, dtype, batch in (("v2", 416, "fp32", 1), ("v2", 416, "fp16", 16), ("v3", 608, "fp16", 32)):
            anchors = bench.COCO_V2 if kind == "v2" else bench.COCO_V3
            names = ["c%d" % i for i in range(80)]
            model = launcher.pick_model(kind)
            net = type(model).create_network(np.reshape(anchors, [-1, 2]), names, False, input_shape=(size, size, 3))
            hg, frac = synth.HEAD_DEFAULTS[kind]
            w = synth.darknet_stream(net, seed=0, num_classes=80, head_gain=hg, obj_bias=0.0)
            cal = launcher.pick_model(kind)         # objectness prior calibrated on the product's own forward: a handful of boxes per image
            cal.build(anchors, names, (size, size, 3), dtype=dtype, max_batch=2, weights=w)
            w = synth.calibrate_model(cal, synth.synthetic_input(2, size, size, 3, seed=999), frac / 4)
            del cal
            wpath = os.path.join(tmp, "%s.weights" % kind)
            base.write_darknet_weights(wpath, w, kind)
            for ext in ("jpg", "png"):
                for pipeline in (False, True):
                    out_dir = os.path.join(tmp, "out_%s_%s_%s_%d" % (kind, dtype, ext, pipeline))
                    params = dict(image_dir=dirs[ext], out_dir=out_dir, batch_size=batch, threshold=0.5, iou_threshold=0.6, anchors=anchors,
                                  class_names=names, input_h=size, input_w=size, input_c=3, checkpoint_path="",
                                  pretrained_weights_path=wpath, cpu_only="False", dtype=dtype, pipeline=str(pipeline), version=kind)
                    yolo = launcher.pick_model(kind)
                    buf = io.StringIO()
                    t0 = time.perf_counter()
                    with contextlib.redirect_stdout(buf):
                        yolo.test(params)
                    wall = time.perf_counter() - t0
                    lines = [l for l in buf.getvalue().splitlines() if ": Found " in l]
                    assert len(lines) == args.images and len(os.listdir(out_dir)) == args.images
                    t = dict(yolo.timing)
                    row = {"net": "%s-%d-%s-b%d" % (kind, size, dtype, batch), "files": ext, "pipeline": pipeline, "images": args.images,
                           "loop_s": round(t.pop("loop_s"), 3), "images_per_s": round(args.images / yolo.timing["loop_s"], 1),
                           "boxes_found": sum(int(l.split("Found ")[1].split(" ")[0]) for l in lines),
                           "wall_incl_build_s": round(wall, 2), "lines_digest": hash(tuple(sorted(l.split(" Saved to ")[0] for l in lines))) & 0xffffffff,
                           "stages_s": {k: round(v, 3) for k, v in t.items() if isinstance(v, float)}}
                    rows.append(row)
                    print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
