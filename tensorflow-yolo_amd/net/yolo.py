"""Driver classes of the HIP backend: `Yolo.test(params)` with the reference's parameter keys,
and the factored-out per-batch body `predict(x_batch)`.

Counterpart of the reference's net/yolo.py (TEST path :41-96; binding classes :198-211).
Training (`train`, `generate_anchors`, loss, batches) is out of scope: those entries raise.
"""
import os

import numpy as np

from .. import _hip
from . import base, dist as ydist, engine, tfckpt, v2, v3


class Yolo(object):
    version = None
    # plug points, bound per version below (same names as the reference)
    create_network = None
    load_weights = None
    find_bounding_boxes = None

    def __init__(self):
        self.net = None
        self.params = None
        self.last_status = None

    # ---- out of scope for an inference backend ------------------------------------------------
    def train(self, params):
        raise NotImplementedError("train mode is not supported by the HIP inference backend")

    def generate_anchors(self, params):
        raise NotImplementedError("anchor mode is not supported by the HIP inference backend")

    def create_loss_fn(self, batch_size, net, anchors, class_names):
        raise NotImplementedError("training is not supported by the HIP inference backend")

    def make_batch(self, net, annotations, batch_size, anchors, class_names, augment_prob):
        raise NotImplementedError("training is not supported by the HIP inference backend")

    def create_train_optimizer(self, loss_fn, learning_rate):
        raise NotImplementedError("training is not supported by the HIP inference backend")

    # ---- network lifecycle ----------------------------------------------------------------------
    def build(self, anchors, class_names, input_shape=(416, 416, 3), dtype="fp32", max_batch=1,
              weights=None, weights_path=None, **engine_kw):
        """create_network + compile for the GPU + load weights (array or Darknet file)."""
        anchors = np.reshape(anchors, [-1, 2])
        self.anchors, self.class_names = anchors, list(class_names)
        net = type(self).create_network(anchors, class_names, False, input_shape=tuple(input_shape))
        net.engine = engine.HipNetwork(net, dtype=dtype, max_batch=max_batch, **engine_kw)
        if self.version != "v3":
            h, w, _ = net[-1].out.hwc
            net.engine.set_head(engine.head_desc_v2(h, w, anchors, len(class_names)))
        if weights is not None:
            v3.attach_weights(net, weights)
        elif weights_path is not None:
            type(self).load_weights(net, weights_path)
        self.net = net
        return net

    def predict(self, x_batch, threshold=0.5, iou_threshold=0.6, nms_mode=_hip.NMS_AGNOSTIC, group=None):
        """One batch: forward + decode + NMS on the GPU (the body of the reference's test loop,
        net/yolo.py:83-86).  x_batch: [B,H,W,C] in [0,1], RGB (NumPy or torch).
        Returns list[B] of list[BoundingBox], each in descending-prob (stable) order.

        Under an initialised `torch.distributed` group of W > 1 ranks (one process per GPU) every rank passes the SAME
        global batch: rank r runs images shard_range(B, r, W), the fixed-size box records are all-gathered (the only
        exchange, net/dist.py) and every rank returns the full list.  `self.last_status` keeps the per-image status
        words (bit 0 candidate overflow, bit 1 more survivors than max_boxes): both raise, the reference has no caps."""
        rank, world = ydist.world(group)
        if world > 1:
            lo, hi = ydist.shard_range(len(x_batch), rank, world)
            return self.predict_shard(x_batch[lo:hi] if hi > lo else None, len(x_batch), threshold, iou_threshold, nms_mode, group)
        return self.predict_shard(x_batch, len(x_batch), threshold, iou_threshold, nms_mode, group)

    def predict_shard(self, x_local, n_global, threshold=0.5, iou_threshold=0.6, nms_mode=_hip.NMS_AGNOSTIC, group=None):
        """predict() for a caller that holds only ITS images of the global batch (Yolo.test under torch.distributed
        preprocesses just the rank's shard): x_local = images shard_range(n_global, rank, world) of the batch, or None
        when the shard is empty.  Returns the list for all n_global images on every rank."""
        eng = self.net.engine
        if not eng.weights_loaded:
            raise RuntimeError("no weights loaded: call load_weights / build(weights=...) first")
        rank, world = ydist.world(group)
        if world > 1:
            n = int(n_global)
            per = -(-n // world)
            if per > eng.max_batch:
                raise ValueError("global batch %d over %d ranks needs max_batch >= %d (engine has %d)" % (n, world, per, eng.max_batch))
            lo, hi = ydist.shard_range(n, rank, world)
            n_local = 0 if x_local is None else len(x_local)
            if n_local != hi - lo:
                raise ValueError("rank %d holds %d images of a global batch of %d, its shard is [%d, %d)" % (rank, n_local, n, lo, hi))
            boxes, counts, status = ydist.detect_sharded(eng, x_local if n_local else None, threshold, iou_threshold, nms_mode, group)
            keep = [r * eng.max_batch + i for r in range(world) for i in range(per)][:n]     # slot of global image g
            status = status.cpu().numpy().reshape(-1)[keep]
            self.last_status = status
            engine.check_status(status)
            lists = ydist.records_to_lists(boxes, counts)
            return base.boxes_from_records([lists[k] for k in keep])
        boxes, counts, status = eng.detect(x_local, threshold, iou_threshold, nms_mode)
        records, self.last_status = engine.records_to_host(boxes, counts, status)
        return base.boxes_from_records(records)

    def forward(self, x_batch):
        """Head logits as a NumPy float32 array in the reference's layout (what sess.run returns)."""
        return self.net.engine.forward(x_batch).cpu().numpy()

    # ---- TEST mode --------------------------------------------------------------------------------
    def test(self, params):
        image_dir = params["image_dir"]
        out_dir = params["out_dir"]
        batch_size = int(params["batch_size"])
        threshold = float(params["threshold"])
        iou_threshold = float(params["iou_threshold"])
        anchors = np.reshape(params["anchors"], [-1, 2])
        class_names = params["class_names"]
        input_shape = (int(params["input_h"]), int(params["input_w"]), int(params["input_c"]))
        checkpoint_path = params.get("checkpoint_path", "")
        pretrained_weights_path = params["pretrained_weights_path"]
        cpu_only = str(params.get("cpu_only", "false")).lower() == "true"
        dtype = params.get("dtype", "fp32")                 # new optional key; fp32 == the reference's arithmetic
        nms_mode = {"agnostic": _hip.NMS_AGNOSTIC, "per_class": _hip.NMS_PER_CLASS}[params.get("nms_mode", "agnostic")]

        image_paths = base.load_image_paths(image_dir)
        if len(image_paths) == 0:
            print("No test images found in {}".format(image_dir))
            return
        if cpu_only:
            print("cpu_only = True is ignored: this backend runs on the MI355X only")

        # one process per GPU (torch.distributed initialised by the launcher, e.g. torchrun): every batch of `batch_size`
        # images shards over the ranks, the box records are all-gathered, rank 0 draws and writes (net/dist.py)
        rank, world = ydist.world()
        # the reference's box lists are unbounded (net/base.py:195-209); the record buffers are not: optional keys raise the caps
        caps = {k: int(params[k]) for k in ("max_boxes", "cand_capacity") if k in params}
        self.build(anchors, class_names, input_shape, dtype=dtype, max_batch=-(-batch_size // world), **caps)
        # same order as the reference (net/yolo.py:71-78, net/base.py:55-61): restore the TensorFlow checkpoint (read without
        # TensorFlow by net/tfckpt.py); when that fails, say so and load the Darknet weights
        restored = False
        if checkpoint_path:
            try:
                v3.attach_weights(self.net, tfckpt.checkpoint_to_darknet(self.net, checkpoint_path))
                restored = True
                print("Checkpoint {} restored.".format(checkpoint_path))
            except Exception as e:
                print("Failed to load {}: {}".format(checkpoint_path, str(e)))
        if not restored:
            type(self).load_weights(self.net, pretrained_weights_path)
            print("Pre-trained weights loaded.")

        if str(params.get("autotune", "false")).lower() == "true":      # new optional key: time every conv tile per layer once, on a
            eng = self.net.engine                                         # full batch of this shape (yolo_net_autotune), instead of the built-in rules
            eng.autotune(np.zeros((eng.max_batch,) + tuple(input_shape), dtype=np.float32))
        # resize / colour order / /255 run on the device with OpenCV's INTER_LINEAR arithmetic (base.preprocess_image_gpu);
        # `preprocess = pillow` (new optional key) keeps the host-side Pillow resampler
        batches = base.generate_test_batch if str(params.get("preprocess", "gpu")).lower() == "pillow" else base.generate_test_batch_gpu
        for start in range(0, len(image_paths), batch_size):
            paths = image_paths[start:start + batch_size]
            # every rank decodes / resizes only ITS shard of the batch (one process per GPU)
            lo, hi = ydist.shard_range(len(paths), rank, world)
            x_local = next(iter(batches(paths[lo:hi], batch_size, input_shape)))[0] if hi > lo else None
            net_boxes = self.predict_shard(x_local, len(paths), threshold, iou_threshold, nms_mode)
            if rank != 0:
                continue
            for boxes, path in zip(net_boxes, paths):
                new_img = base.draw_boxes(path, boxes, class_names)
                file_name, file_ext = os.path.splitext(os.path.basename(path))
                out_path = os.path.join(out_dir, "{}_out{}".format(file_name, file_ext))
                base.save_image(new_img, out_path)
                print("{}: Found {} objects. Saved to {}".format(file_name, len(boxes), out_path))
        if rank == 0:
            print("Done")


class YoloV2(Yolo):
    version = "v2"
    create_network = staticmethod(v2.create_full_network)
    load_weights = staticmethod(v2.load_weights)
    find_bounding_boxes = staticmethod(v2.find_bounding_boxes)


class YoloV2Tiny(Yolo):
    """Not in the reference (it ships only the tiny-voc anchors); same plug points."""
    version = "v2-tiny"
    create_network = staticmethod(v2.create_tiny_network)
    load_weights = staticmethod(v2.load_weights)
    find_bounding_boxes = staticmethod(v2.find_bounding_boxes)


class YoloV3(Yolo):
    version = "v3"
    create_network = staticmethod(v3.create_network)
    load_weights = staticmethod(v3.load_weights)
    find_bounding_boxes = staticmethod(v3.find_bounding_boxes)
