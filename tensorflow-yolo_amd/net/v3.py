"""YOLOv3 (Darknet-53 + three-scale head) for the HIP backend.

Same three plug points as the reference's net/v3.py -- `create_network` (:9-94), `load_weights`
(:98-106), `find_bounding_boxes` (:140-151) -- with the same signatures; the layer list has the
same 109 entries in the same order, so layer indices (route sources 62 and 37) and the Darknet
weight order carry over.
"""
import numpy as np

from .. import _hip
from . import base, engine
from .layers import conv2d_bn_act, detection_layer, input_layer, route, shortcut, upsample, yolo_layer

# Darknet-53 trunk: (filters of the stride-2 conv, number of residual blocks that follow)
_STAGES = ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4))
_SKIP_FINE = 36 + 1     # output of the last 256-channel block (+1: the input layer is entry 0)
_SKIP_MID = 61 + 1      # output of the last 512-channel block


class Network(list):
    """The layer list plus what the runtime attaches to it."""
    version = "v3"
    engine = None
    darknet_weights = None


def create_network(anchors, class_names, is_training, scope="yolo", input_shape=(416, 416, 3)):
    num_classes = len(class_names)
    per_scale = np.reshape(anchors, [3, -1, 2])[::-1, :, :]     # coarsest head gets the largest anchors
    conv2d_bn_act.reset()
    net = Network()
    kw = dict(is_training=is_training, scope=scope)

    def conv(filters, ksize, stride=1, **extra):
        net.append(conv2d_bn_act(net[-1].out, filters, ksize, stride, **dict(kw, **extra)))

    def residual(filters):
        block_in = net[-1]
        conv(filters // 2, 1)
        conv(filters, 3)
        net.append(shortcut(net[-1].out, block_in.out))

    def head(filters, sub_anchors):
        for _ in range(3):
            conv(filters, 1)
            conv(filters * 2, 3)
        conv(len(sub_anchors) * (5 + num_classes), 1, 1, use_batch_normalization=False, activation_fn="linear")
        net.append(yolo_layer(net[-1].out, sub_anchors, num_classes, input_shape))
        return net[-1]

    def lateral(filters, skip_index):
        net.append(route([net[-4].out]))            # the 1x1 output two convs before the head conv
        conv(filters, 1)
        net.append(upsample(net[-1].out, 2))
        net.append(route([net[-1].out, net[skip_index].out]))

    net.append(input_layer([None, input_shape[0], input_shape[1], input_shape[2]], "input"))
    conv(32, 3)
    for filters, blocks in _STAGES:
        conv(filters, 3, 2)
        for _ in range(blocks):
            residual(filters)

    yolos = [head(512, per_scale[0])]
    lateral(256, _SKIP_MID)
    yolos.append(head(256, per_scale[1]))
    lateral(128, _SKIP_FINE)
    yolos.append(head(128, per_scale[2]))
    net.append(detection_layer(yolos))
    return net


def load_weights(layers, weights_path):
    """Reads the Darknet file and hands the float stream to the network's engine (uploaded eagerly;
    nothing is returned to run, unlike the reference's list of tf.assign ops)."""
    print("Reading pre-trained weights from {}".format(weights_path))
    header, weights = base.read_darknet_weights(weights_path, "v3")
    print("{} {} {} {} {}".format(*header))
    print("Found {} weight values.".format(len(weights)))
    return attach_weights(layers, weights)


def attach_weights(layers, weights):
    need = sum(l.weight_count() for l in layers if isinstance(l, conv2d_bn_act))
    if need != len(weights):
        # stricter than the reference, which only prints the two counts (net/base.py:44)
        raise ValueError("weight file holds {} values, the network needs {}".format(len(weights), need))
    layers.darknet_weights = np.ascontiguousarray(weights, dtype=np.float32)
    if layers.engine is not None:
        layers.engine.load_weights(layers.darknet_weights)
    print("Weights ready ({}/{} read)".format(need, len(weights)))
    return []


def find_bounding_boxes(net_out, net, threshold, iou_threshold, anchors, class_names, nms_mode=0):
    """Head decode over the three scales + one NMS per image, on the GPU (libyolo_hip
    yolo_decode_nms).  net_out: [B, rows, 5+C] NumPy array or torch device tensor."""
    head = engine.head_desc_v3(net[-1].yolos)
    eng = getattr(net, "engine", None)
    records, _ = engine.decode_nms(head, net_out, threshold, iou_threshold, nms_mode,
                                   cand_capacity=eng.cand_capacity if eng else 4096,
                                   max_boxes=eng.max_boxes if eng else _hip.DEFAULT_MAX_BOXES)
    return base.boxes_from_records(records)
