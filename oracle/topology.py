"""Oracle (test infrastructure): layer lists of the reference networks, restated as data.

Each entry is a tuple whose first element is the op name; list index == position
in the reference's ``layers`` list (0 = input layer), so the route sources quoted
in SURVEY App. B can be asserted directly.

  ("input", h, w, c)
  ("conv", src, filters, ksize, stride, bn, act)      net/layers.py:17-67
  ("maxpool", src, ksize, stride)                     net/layers.py:70-81
  ("route", [srcs])                                   net/layers.py:84-87
  ("reorg", src, stride)                              net/layers.py:90-97
  ("shortcut", src, skip)                             net/layers.py:100-103
  ("upsample", src, stride)                           net/layers.py:112-116
  ("yolo", src, anchors_px[(w,h)..])                  net/layers.py:126-134
  ("detection", [yolo idxs])                          net/layers.py:119-123

Follows net/v2.py:11-60 (YOLOv2), net/v3.py:9-94 (YOLOv3).  tiny-YOLOv2 is NOT in
the reference: it is the upstream Darknet yolov2-tiny-voc.cfg
expressed in the reference's layer vocabulary.
"""


def yolov2(num_anchors, num_classes, input_shape=(416, 416, 3)):
    L = [("input",) + tuple(input_shape)]

    def conv(f, k, s=1, bn=True, act="leaky", src=None):
        L.append(("conv", len(L) - 1 if src is None else src, f, k, s, bn, act))

    def pool():
        L.append(("maxpool", len(L) - 1, 2, 2))

    for f in (32, 64):                      # net/v2.py:20-22
        conv(f, 3)
        pool()
    for f in (128, 256):                    # net/v2.py:24-29
        conv(f, 3)
        conv(f // 2, 1)
        conv(f, 3)
        pool()
    for f, k in ((512, 3), (256, 1), (512, 3), (256, 1), (512, 3)):   # :31-35
        conv(f, k)
    pool()                                  # :36
    for f, k in ((1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)):  # :38-42
        conv(f, k)
    conv(1024, 3)                           # :44
    conv(1024, 3)                           # :45
    L.append(("route", [len(L) - 9]))       # :46  layers[-9]
    conv(64, 1)                             # :47
    L.append(("reorg", len(L) - 1, 2))      # :48
    L.append(("route", [len(L) - 1, len(L) - 4]))   # :49
    conv(1024, 3)                           # :50
    conv(num_anchors * (5 + num_classes), 1, 1, bn=False, act="linear")  # :52-56
    return L


def yolov3(anchors_px, num_classes, input_shape=(416, 416, 3)):
    """anchors_px: flat or [9,2] list in pixels, ini order (small -> large)."""
    import numpy as np
    anc = np.reshape(np.asarray(anchors_px), [3, -1, 2])[::-1, :, :]    # net/v3.py:11
    L = [("input",) + tuple(input_shape)]

    def conv(f, k, s=1, bn=True, act="leaky"):
        L.append(("conv", len(L) - 1, f, k, s, bn, act))

    def block(f):                           # net/v3.py:16-19
        conv(f, 1)
        conv(2 * f, 3)
        L.append(("shortcut", len(L) - 1, len(L) - 3))

    conv(32, 3)
    conv(64, 3, 2)
    block(32)
    conv(128, 3, 2)
    for _ in range(2):
        block(64)
    conv(256, 3, 2)
    for _ in range(8):
        block(128)
    conv(512, 3, 2)
    for _ in range(8):
        block(256)
    conv(1024, 3, 2)
    for _ in range(4):
        block(512)

    def head(f, a):
        for _ in range(3):
            conv(f, 1)
            conv(2 * f, 3)
        conv(len(a) * (5 + num_classes), 1, 1, bn=False, act="linear")
        L.append(("yolo", len(L) - 1, [tuple(x) for x in a.tolist()]))
        return len(L) - 1

    y1 = head(512, anc[0])                  # net/v3.py:46-55
    L.append(("route", [len(L) - 4]))       # :56
    conv(256, 1)
    L.append(("upsample", len(L) - 1, 2))
    L.append(("route", [len(L) - 1, 61 + 1]))   # :59
    y2 = head(256, anc[1])
    L.append(("route", [len(L) - 4]))       # :72
    conv(128, 1)
    L.append(("upsample", len(L) - 1, 2))
    L.append(("route", [len(L) - 1, 36 + 1]))   # :75
    y3 = head(128, anc[2])
    L.append(("detection", [y1, y2, y3]))   # :90
    return L


def tiny_yolov2(num_anchors, num_classes, input_shape=(416, 416, 3)):
    """Upstream Darknet yolov2-tiny-voc.cfg; NOT in the reference."""
    L = [("input",) + tuple(input_shape)]
    for f in (16, 32, 64, 128, 256):
        L.append(("conv", len(L) - 1, f, 3, 1, True, "leaky"))
        L.append(("maxpool", len(L) - 1, 2, 2))
    L.append(("conv", len(L) - 1, 512, 3, 1, True, "leaky"))
    L.append(("maxpool", len(L) - 1, 2, 1))
    L.append(("conv", len(L) - 1, 1024, 3, 1, True, "leaky"))
    L.append(("conv", len(L) - 1, 1024, 3, 1, True, "leaky"))
    L.append(("conv", len(L) - 1, num_anchors * (5 + num_classes), 1, 1, False, "linear"))
    return L


def shapes(L):
    """Output (h, w, c) per layer, following the TF shape rules of net/layers.py."""
    S = []
    for op in L:
        k = op[0]
        if k == "input":
            S.append((op[1], op[2], op[3]))
        elif k == "conv":
            h, w, _ = S[op[1]]
            s = op[4]
            # stride 1 -> SAME; stride>1 -> explicit pad (k-1) then VALID  (layers.py:28-30)
            S.append((h, w, op[2]) if s == 1 else
                     ((h + op[3] - 1 - op[3]) // s + 1, (w + op[3] - 1 - op[3]) // s + 1, op[2]))
        elif k == "maxpool":
            h, w, c = S[op[1]]
            ks, s = op[2], op[3]
            S.append((h, w, c) if s == 1 else ((h + ks - 1 - ks) // s + 1, (w + ks - 1 - ks) // s + 1, c))
        elif k == "route":
            h, w, _ = S[op[1][0]]
            S.append((h, w, sum(S[i][2] for i in op[1])))
        elif k == "reorg":
            h, w, c = S[op[1]]
            S.append((h // op[2], w // op[2], c * op[2] * op[2]))
        elif k == "shortcut":
            S.append(S[op[1]])
        elif k == "upsample":
            h, w, c = S[op[1]]
            S.append((h * op[2], w * op[2], c))
        elif k == "yolo":
            S.append(S[op[1]])
        elif k == "detection":
            S.append(None)
        else:
            raise ValueError(k)
    return S


def conv_weight_count(L):
    """Number of float32 values the Darknet stream must hold (net/base.py:26-46)."""
    S = shapes(L)
    n = 0
    for op in L:
        if op[0] == "conv":
            cin = S[op[1]][2]
            n += op[2] * cin * op[3] * op[3] + (4 * op[2] if op[5] else op[2])
    return n


def conv_flops(L):
    """2*Ho*Wo*Cout*k*k*Cin summed over convs."""
    S = shapes(L)
    f = 0
    for i, op in enumerate(L):
        if op[0] == "conv":
            h, w, c = S[i]
            f += 2 * h * w * c * op[3] * op[3] * S[op[1]][2]
    return f
