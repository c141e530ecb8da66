"""Oracle (test infrastructure): seeded synthetic head tensors shared by the fixture
generator (oracle/gen_golden.py, runs the REFERENCE) and the tests (run the oracle /
the HIP path).  Inputs are regenerated from the seed; only outputs are committed.

Value distributions follow SURVEY 8d: randn*scale with the objectness logit shifted so
that tens to hundreds of candidates pass; plus engineered ties (equal prob) and an
empty case.
"""
import numpy as np

COCO_V2_ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434,
                   7.88282, 3.52778, 9.77052, 9.16828]            # config/yolo_2.ini:52
VOC_TINY_ANCHORS = [1.08, 1.19, 3.42, 4.41, 6.63, 11.38, 9.42, 5.11, 16.62, 10.52]   # resource/yolov2-tiny-voc.anchors:1
COCO_V3_ANCHORS = [10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119,
                   116, 90, 156, 198, 373, 326]                    # config/yolo_3.ini:39

# name -> dict(version, input, batch, classes, seed, scale, obj_shift, threshold, iou, anchors, ties)
CASES = {
    "v2_416_thr01": dict(version=2, input=416, batch=2, classes=80, seed=11, scale=2.0, obj_shift=0.0,
                         cls_boost=0.0, threshold=0.1, iou=0.6, anchors=COCO_V2_ANCHORS, ties=0),
    "v2_416_thr05": dict(version=2, input=416, batch=2, classes=80, seed=12, scale=1.0, obj_shift=1.0,
                         cls_boost=6.0, threshold=0.5, iou=0.6, anchors=COCO_V2_ANCHORS, ties=6),
    "v2_tinyvoc_416": dict(version=2, input=416, batch=1, classes=20, seed=13, scale=1.5, obj_shift=0.5,
                           cls_boost=4.0, threshold=0.3, iou=0.5, anchors=VOC_TINY_ANCHORS, ties=0),
    "v2_416_empty": dict(version=2, input=416, batch=1, classes=80, seed=14, scale=1.0, obj_shift=-8.0,
                         cls_boost=0.0, threshold=0.5, iou=0.6, anchors=COCO_V2_ANCHORS, ties=0),
    "v3_416": dict(version=3, input=416, batch=2, classes=80, seed=21, scale=2.0, obj_shift=-4.0,
                   cls_boost=0.0, threshold=0.5, iou=0.6, anchors=COCO_V3_ANCHORS, ties=8),
    "v3_608": dict(version=3, input=608, batch=1, classes=80, seed=22, scale=2.0, obj_shift=-5.0,
                   cls_boost=0.0, threshold=0.5, iou=0.6, anchors=COCO_V3_ANCHORS, ties=0),
    "v3_416_empty": dict(version=3, input=416, batch=1, classes=80, seed=23, scale=1.0, obj_shift=-9.0,
                         cls_boost=0.0, threshold=0.5, iou=0.6, anchors=COCO_V3_ANCHORS, ties=0),
    "v3_320_lowthr": dict(version=3, input=320, batch=1, classes=80, seed=24, scale=2.0, obj_shift=-3.0,
                          cls_boost=0.0, threshold=0.25, iou=0.45, anchors=COCO_V3_ANCHORS, ties=4),
    # the headline size WITH suppression work (v3_608 above keeps all of its 141 candidates): box logits scaled down
    # so that boxes are anchor-sized around their cell centres and neighbouring cells / anchors overlap
    "v3_608_dense": dict(version=3, input=608, batch=1, classes=80, seed=25, scale=2.0, obj_shift=-3.9,
                         cls_boost=0.0, threshold=0.5, iou=0.3, anchors=COCO_V3_ANCHORS, ties=3, box_scale=0.1),
}


def head_rows(case):
    """Number of (cell, anchor) rows per image and the per-row width."""
    c = CASES[case] if isinstance(case, str) else case
    g = c["input"] // 32
    if c["version"] == 2:
        return g * g * (len(c["anchors"]) // 2), 5 + c["classes"]
    return (g * g + 4 * g * g + 16 * g * g) * 3, 5 + c["classes"]


def make_head(case):
    """Seeded head tensor in the reference's layout: v2 [B,g,g,A*(5+C)], v3 [B,rows,5+C]."""
    c = CASES[case] if isinstance(case, str) else case
    rng = np.random.RandomState(c["seed"])
    rows, width = head_rows(c)
    t = (rng.randn(c["batch"], rows, width) * c["scale"]).astype(np.float32)
    t[..., 4] += np.float32(c["obj_shift"])
    if c.get("box_scale"):
        t[..., 0:4] *= np.float32(c["box_scale"])
    if c["cls_boost"]:
        hot = rng.randint(0, c["classes"], size=(c["batch"], rows))
        bi, ri = np.meshgrid(np.arange(c["batch"]), np.arange(rows), indexing="ij")
        t[bi, ri, 5 + hot] += np.float32(c["cls_boost"])
    if c["ties"]:
        # engineered ties: copy the objectness (and for v2 the class logits) of one passing
        # row onto a few others so that several boxes carry bit-identical prob
        for b in range(c["batch"]):
            src = int(np.argmax(t[b, :, 4]))
            dst = rng.choice(rows, size=c["ties"], replace=False)
            for d in dst:
                t[b, d, 4:] = t[b, src, 4:]
    if c["version"] == 2:
        g = c["input"] // 32
        return t.reshape(c["batch"], g, g, -1)
    return t


# engineered NMS-only cases: list of (x, y, w, h, class_idx, prob), iou threshold
NMS_CASES = {
    "iou_exactly_at_threshold": (
        [(0.5, 0.5, 1.0, 1.0, 0, 0.9), (0.5, 0.25, 1.0, 0.5, 1, 0.8), (3.0, 3.0, 0.5, 0.5, 2, 0.7)], 0.5),
    "iou_just_below_threshold": (
        [(0.5, 0.5, 1.0, 1.0, 0, 0.9), (0.5, 0.25, 1.0, 0.5, 1, 0.8)], 0.5000001),
    "equal_prob_keeps_scan_order": (
        [(0.2, 0.2, 0.1, 0.1, 3, 0.75), (0.8, 0.8, 0.1, 0.1, 4, 0.75), (0.2, 0.2, 0.1, 0.1, 5, 0.75),
         (0.5, 0.5, 0.1, 0.1, 6, 0.75)], 0.6),
    "zero_area_union_floor": (
        [(0.5, 0.5, 0.0, 0.0, 0, 0.9), (0.5, 0.5, 0.0, 0.0, 1, 0.8)], 0.6),
    "chain_suppression": (
        [(0.50, 0.5, 0.4, 0.4, 0, 0.95), (0.60, 0.5, 0.4, 0.4, 0, 0.90), (0.70, 0.5, 0.4, 0.4, 0, 0.85),
         (0.80, 0.5, 0.4, 0.4, 0, 0.80), (0.90, 0.5, 0.4, 0.4, 0, 0.75)], 0.5),
    "single": ([(0.3, 0.3, 0.2, 0.2, 7, 0.6)], 0.6),
    "empty": ([], 0.6),
}
