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
        pillow = str(params.get("preprocess", "gpu")).lower() == "pillow"
        # One process: the loop is a three-stage pipeline (new optional key `pipeline`, default True): worker threads decode the files
        # of batch i + 1 and draw / encode / write the images of batch i - 1 while the GPU runs batch i; the box records come back by
        # an asynchronous copy into pinned memory.  Same files, same console lines in the same order as the serial loop.
        import time
        t_loop = time.perf_counter()
        self.timing = {"images": len(image_paths), "batch_size": batch_size}      # (seconds per stage of the loop: tools/e2e_launcher.py)
        if world == 1 and not pillow and str(params.get("pipeline", "true")).lower() == "true":
            self.timing["mode"] = "pipelined"
            self._test_pipelined(image_paths, out_dir, batch_size, input_shape, threshold, iou_threshold, nms_mode, class_names,
                                 workers=int(params.get("workers", 0)), timings=self.timing)
            self.timing["loop_s"] = time.perf_counter() - t_loop
            print("Done")
            return
        self.timing.update(mode="serial", decode_preprocess=0.0, predict=0.0, draw_save=0.0)
        batches = base.generate_test_batch if pillow else base.generate_test_batch_gpu
        for start in range(0, len(image_paths), batch_size):
            paths = image_paths[start:start + batch_size]
            # every rank decodes / resizes only ITS shard of the batch (one process per GPU)
            lo, hi = ydist.shard_range(len(paths), rank, world)
            t0 = time.perf_counter()
            x_local = next(iter(batches(paths[lo:hi], batch_size, input_shape)))[0] if hi > lo else None
            t1 = time.perf_counter()
            net_boxes = self.predict_shard(x_local, len(paths), threshold, iou_threshold, nms_mode)
            t2 = time.perf_counter()
            self.timing["decode_preprocess"] += t1 - t0
            self.timing["predict"] += t2 - t1
            if rank != 0:
                continue
            for boxes, path in zip(net_boxes, paths):
                new_img = base.draw_boxes(path, boxes, class_names)
                file_name, file_ext = os.path.splitext(os.path.basename(path))
                out_path = os.path.join(out_dir, "{}_out{}".format(file_name, file_ext))
                base.save_image(new_img, out_path)
                print("{}: Found {} objects. Saved to {}".format(file_name, len(boxes), out_path))
            self.timing["draw_save"] += time.perf_counter() - t2
        self.timing["loop_s"] = time.perf_counter() - t_loop
        if rank == 0:
            print("Done")

    def _test_pipelined(self, image_paths, out_dir, batch_size, input_shape, threshold, iou_threshold, nms_mode, class_names, workers=0,
                        timings=None):
        """The body of the reference's test loop (net/yolo.py:80-95) as a pipeline over batches:

            worker threads   decode_image() of the files of batch i + 1 (Pillow releases the GIL inside its codecs)
            this thread      batch i: uint8 pixels -> pinned staging -> device (async), yolo_preprocess_resize of every image
                             straight into the batch tensor, yolo_net_detect, async copy of the record buffer
                             [counts | status | boxes] to pinned memory + an event -- nothing here waits for the GPU
            worker PROCESSES batch i - 1, once its event has fired: records -> BoundingBox lists, draw_boxes on the pixels
                             already decoded (no second read of the file), encode + write `<stem>_out<ext>`
                             (base.draw_save_task; drawing holds the GIL, so threads only for directories of < 64 files)

        The console lines are printed by this thread in image order, a batch's lines when its files are on disk.
        timings: optional dict that receives the summed seconds per stage (tools/e2e_launcher.py)."""
        import time
        from concurrent.futures import ThreadPoolExecutor
        import torch
        eng = self.net.engine
        torch_dev = eng.device
        lib = _hip.lib()
        h, w, c = (int(v) for v in input_shape)
        if workers <= 0:
            workers = max(2, min(16, (os.cpu_count() or 4) - 2))
        t = timings if timings is not None else {}
        for k in ("decode_wait", "upload_resize_enqueue", "detect_enqueue", "records_wait", "boxes", "draw_save_wait"):
            t.setdefault(k, 0.0)
        batches = [image_paths[i:i + batch_size] for i in range(0, len(image_paths), batch_size)]
        pool = ThreadPoolExecutor(max_workers=min(8, workers))          # decoders (Pillow's codecs release the GIL)
        # drawing + encoding hold the GIL, so the writers are PROCESSES (spawned: they import PIL / NumPy / net.base, never torch,
        # never the GPU); a handful of files is not worth their start-up: threads then
        if len(image_paths) >= 64:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            writers = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"))
            for _ in range(workers):        # (the processes start now, beside the first batches, not at the first file to write)
                writers.submit(int)
        else:
            writers = pool
        decode = lambda paths: [pool.submit(base.decode_image, p) for p in paths]
        stream = torch.cuda.current_stream(torch_dev)
        # two sets of per-batch resources, used alternately: batch i + 1 is prepared while batch i's records are still in flight
        x_dev = [torch.empty((batch_size, h, w, c), dtype=torch.float32, device=torch_dev) for _ in range(2)]
        rec_host = [torch.empty(ydist.record_words(eng.max_batch, eng.max_boxes), dtype=torch.int32).pin_memory() for _ in range(2)]
        stage_host, stage_dev = [None, None], [None, None]
        events = [torch.cuda.Event() for _ in range(2)]

        def finish(job):
            """batch whose GPU work was enqueued earlier: wait for its records, hand the images to the writers"""
            slot, paths, rgbs = job
            t0 = time.perf_counter()
            events[slot].synchronize()
            t["records_wait"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            n = len(paths)
            boxes, counts, status = ydist.split_records(rec_host[slot], eng.max_batch, eng.max_boxes)
            records, self.last_status = engine.records_to_host(boxes[:n], counts[:n], status[:n])
            t["boxes"] += time.perf_counter() - t0
            return [writers.submit(base.draw_save_task, p, rgb, r, list(class_names), out_dir) for p, rgb, r in zip(paths, rgbs, records)]

        from collections import deque
        written = deque()               # futures of the writers, in image order
        max_pending = 4 * workers + 2 * batch_size

        def flush(everything=False):
            """print the lines of the files that are on disk, in image order; wait only when too many writes are queued (or at the end)"""
            t0 = time.perf_counter()
            while written and (everything or len(written) > max_pending or written[0].done()):
                print(written.popleft().result())
            t["draw_save_wait"] += time.perf_counter() - t0

        os.makedirs(out_dir, exist_ok=True)
        ahead = max(2 * batch_size, 32)         # images whose decode is requested ahead of the batch the GPU is given
        decoding, nxt, in_flight = deque(), 0, 0
        job = None
        for bi, paths in enumerate(batches):
            slot = bi & 1
            while nxt < len(batches) and (nxt <= bi or in_flight + len(batches[nxt]) <= ahead):
                decoding.append(decode(batches[nxt]))
                in_flight += len(batches[nxt])
                nxt += 1
            t0 = time.perf_counter()
            rgbs = [f.result() for f in decoding.popleft()]
            in_flight -= len(paths)
            t["decode_wait"] += time.perf_counter() - t0
            for p, rgb in zip(paths, rgbs):
                if rgb is None:
                    raise IOError("cannot read image {}".format(p))
            # ---- batch bi on the GPU (enqueue only) -----------------------------------------------------------------------------------
            t0 = time.perf_counter()
            total = sum((int(r.size) + 255) // 256 * 256 for r in rgbs)     # (every image starts on a 256-byte boundary)
            if stage_host[slot] is None or stage_host[slot].numel() < total:
                stage_host[slot] = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8).pin_memory()
                stage_dev[slot] = torch.empty(stage_host[slot].numel(), dtype=torch.uint8, device=torch_dev)
            off = 0
            spans = []
            host_np = stage_host[slot].numpy()
            for rgb in rgbs:
                n = int(rgb.size)
                host_np[off:off + n] = rgb.reshape(-1)
                spans.append((off, rgb.shape[0], rgb.shape[1]))
                off += (n + 255) // 256 * 256
            stage_dev[slot][:off].copy_(stage_host[slot][:off], non_blocking=True)
            x = x_dev[slot][:len(paths)]
            for i, (o, ih, iw) in enumerate(spans):
                _hip.check(lib.yolo_preprocess_resize(stage_dev[slot].data_ptr() + o, ih, iw, iw * 3, x[i].data_ptr(), h, w, 0,
                                                      stream.cuda_stream), "yolo_preprocess_resize")
            t["upload_resize_enqueue"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            eng.detect(x, threshold, iou_threshold, nms_mode)
            rec_host[slot].copy_(eng.records, non_blocking=True)
            events[slot].record(stream)
            t["detect_enqueue"] += time.perf_counter() - t0
            # ---- batch bi - 1: records -> boxes -> writers; the lines of whatever is on disk by now ------------------------------------
            if job is not None:
                written.extend(finish(job))
            flush()
            job = (slot, paths, rgbs)
        if job is not None:
            written.extend(finish(job))
        flush(everything=True)
        pool.shutdown()
        if writers is not pool:
            writers.shutdown()


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
