"""YOLOv2 (Darknet-19 + passthrough) and tiny-YOLOv2 for the HIP backend.

Plug points with the reference's names and signatures: `create_full_network` (reference
net/v2.py:11-60), `load_weights` (:63-79), `find_bounding_boxes` (:83-90).  The layer list has the
same 32 entries in the same order (route sources resolve to entries 17 and 25; Darknet weight
order unchanged).  `create_tiny_network` is NOT in the reference: it is upstream Darknet's
yolov2-tiny-voc.cfg written in the same vocabulary (the reference ships only its anchors,
resource/yolov2-tiny-voc.anchors).

Training pieces of the reference file (loss, ground truth, batches, anchor k-means, :123-323)
are out of scope for this inference backend.
"""
import numpy as np

from .. import _hip
from . import base, engine
from .layers import conv2d_bn_act, input_layer, max_pool2d, reorg, route
from .v3 import Network as _Network, attach_weights

# Darknet-19 trunk: numbers are (filters, ksize); "M" is a 2x2/2 max-pool
_TRUNK = ((32, 3), "M", (64, 3), "M",
          (128, 3), (64, 1), (128, 3), "M",
          (256, 3), (128, 1), (256, 3), "M",
          (512, 3), (256, 1), (512, 3), (256, 1), (512, 3), "M",
          (1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3),
          (1024, 3), (1024, 3))
_PASSTHROUGH_FROM = 17      # the 26x26x512 map (last conv before the fifth pool)


class Network(_Network):
    version = "v2"
    anchors = None
    num_classes = None


def _start(input_shape):
    conv2d_bn_act.reset()
    net = Network()
    net.append(input_layer([None, input_shape[0], input_shape[1], input_shape[2]], "input"))
    return net


def _head_conv(net, num_anchors, num_classes, kw):
    net.append(conv2d_bn_act(net[-1].out, num_anchors * (5 + num_classes), 1, 1,
                             use_batch_normalization=False, activation_fn="linear", **kw))


def create_full_network(anchors, class_names, is_training, scope="yolo", input_shape=(416, 416, 3)):
    num_anchors, num_classes = len(anchors), len(class_names)
    kw = dict(is_training=is_training, scope=scope)
    net = _start(input_shape)
    for item in _TRUNK:
        if item == "M":
            net.append(max_pool2d(net[-1].out, 2, stride=2))
        else:
            net.append(conv2d_bn_act(net[-1].out, item[0], item[1], stride=1, **kw))
    trunk_out = net[-1]
    net.append(route([net[_PASSTHROUGH_FROM].out]))
    net.append(conv2d_bn_act(net[-1].out, 64, 1, stride=1, **kw))
    net.append(reorg(net[-1].out, 2))
    net.append(route([net[-1].out, trunk_out.out]))
    net.append(conv2d_bn_act(net[-1].out, 1024, 3, stride=1, **kw))
    _head_conv(net, num_anchors, num_classes, kw)
    net.anchors, net.num_classes = np.reshape(np.asarray(anchors, dtype=np.float64), [-1, 2]), num_classes
    return net


def create_tiny_network(anchors, class_names, is_training, scope="yolo", input_shape=(416, 416, 3)):
    """tiny-YOLOv2 (9 convs): 16-32-64-128-256 with 2x2/2 pools, 512 + 2x2/1 pool, 1024, 1024, head."""
    num_anchors, num_classes = len(anchors), len(class_names)
    kw = dict(is_training=is_training, scope=scope)
    net = _start(input_shape)
    for filters in (16, 32, 64, 128, 256):
        net.append(conv2d_bn_act(net[-1].out, filters, 3, stride=1, **kw))
        net.append(max_pool2d(net[-1].out, 2, stride=2))
    net.append(conv2d_bn_act(net[-1].out, 512, 3, stride=1, **kw))
    net.append(max_pool2d(net[-1].out, 2, stride=1))
    net.append(conv2d_bn_act(net[-1].out, 1024, 3, stride=1, **kw))
    net.append(conv2d_bn_act(net[-1].out, 1024, 3, stride=1, **kw))
    _head_conv(net, num_anchors, num_classes, kw)
    net.anchors, net.num_classes = np.reshape(np.asarray(anchors, dtype=np.float64), [-1, 2]), num_classes
    return net


def load_weights(layers, weights_path):
    print("Reading pre-trained weights from {}".format(weights_path))
    header, weights = base.read_darknet_weights(weights_path, "v2")
    print("major, minor, revision: {}, {}, {}".format(*header[:3]))
    print("SEEN: ", header[3])
    print("Found {} weight values.".format(len(weights)))
    return attach_weights(layers, weights)


def find_bounding_boxes(net_out, net, threshold, iou_threshold, anchors, class_names, nms_mode=0):
    """Decode (p = sigmoid(obj) * max softmax(cls)) + NMS per image on the GPU.
    net_out: [B, h, w, A*(5+C)] NumPy array or torch device tensor; `net` may be None as in the
    reference, which never reads it on this path."""
    anchors = np.reshape(np.asarray(anchors, dtype=np.float64), [-1, 2])
    h, w = int(net_out.shape[1]), int(net_out.shape[2])
    head = engine.head_desc_v2(h, w, anchors, len(class_names))
    eng = getattr(net, "engine", None)
    records, _ = engine.decode_nms(head, net_out, threshold, iou_threshold, nms_mode,
                                   cand_capacity=eng.cand_capacity if eng else 4096,
                                   max_boxes=eng.max_boxes if eng else _hip.DEFAULT_MAX_BOXES)
    return base.boxes_from_records(records)
