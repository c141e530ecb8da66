"""Shared helpers of the HIP backend: result type, Darknet weight-file reading, NMS entry,
image listing / preprocessing / drawing.

Counterpart of the reference's net/base.py; the arithmetic that file does in NumPy/Python
(sigmoid/softmax/IoU/NMS, :171-209) runs here in libyolo_hip kernels instead.
"""
import ctypes as C
import os

import numpy as np

from .. import _hip

COLORS = [(0, 0, 255), (0, 255, 0), (255, 0, 0), (0, 255, 255), (255, 255, 0), (255, 0, 255)]    # BGR, as the reference draws
IMAGE_EXTENSIONS = ("jpg", "bmp", "png", "gif")


class BoundingBox(object):
    """Same fields and helpers as the reference's result type (net/base.py:257-272)."""

    __slots__ = ("x", "y", "w", "h", "cx", "cy", "class_idx", "prob")     # (a batch of YOLOv3-608 yields ~3 000 of these)

    def __init__(self, x=0., y=0., w=0., h=0., cx=0, cy=0, class_idx=-1, prob=-1.):
        self.x = x; self.y = y; self.w = w; self.h = h
        self.cx = cx; self.cy = cy
        self.class_idx = class_idx
        self.prob = prob

    def get_top_left(self, h=1., w=1.):
        return (self.x - self.w / 2.) * w, (self.y - self.h / 2.) * h

    def get_bottom_right(self, h=1., w=1.):
        return (self.x + self.w / 2.) * w, (self.y + self.h / 2.) * h

    def __repr__(self):
        return "BoundingBox(x=%.4f, y=%.4f, w=%.4f, h=%.4f, class_idx=%d, prob=%.4f)" % (
            self.x, self.y, self.w, self.h, self.class_idx, self.prob)


def boxes_from_records(records):
    return [[BoundingBox(r[0], r[1], r[2], r[3], 0, 0, r[4], r[5]) for r in img] for img in records]


def non_maximum_suppression(boxes, iou_threshold, per_class=False):
    """Greedy NMS of a host list of BoundingBox on the GPU (yolo_nms_host).
    Semantics of the reference (net/base.py:195-209): stable sort by prob descending, suppress
    when IoU >= threshold, class-agnostic unless per_class."""
    n = len(boxes)
    if n == 0:
        return []
    xywh = np.array([[b.x, b.y, b.w, b.h] for b in boxes], dtype=np.float64)
    prob = np.array([b.prob for b in boxes], dtype=np.float32)
    cls = np.array([b.class_idx for b in boxes], dtype=np.int32)
    keep = np.zeros(n, dtype=np.int32)
    n_keep = C.c_int32(0)
    lib = _hip.lib()
    _hip.check(lib.yolo_nms_host(xywh.ctypes.data, prob.ctypes.data, cls.ctypes.data, n, float(iou_threshold),
                                 _hip.NMS_PER_CLASS if per_class else _hip.NMS_AGNOSTIC, keep.ctypes.data,
                                 C.byref(n_keep)), "yolo_nms_host")
    return [boxes[i] for i in keep[:n_keep.value]]


# ---- Darknet .weights files ----------------------------------------------------------------------
def read_darknet_weights(weights_path, version):
    """Header + float32 body.  v2 files: int32 major, minor, revision and a 4-byte `seen`
    (reference net/v2.py:69-75 reads 4 bytes for both header generations); v3 files: five int32
    (reference net/v3.py:102).  Returns (header tuple, float32 array)."""
    with open(weights_path, "rb") as f:
        if version == "v3":
            header = tuple(int(v) for v in np.fromfile(f, count=5, dtype=np.int32))
        else:
            major, minor, revision = (int(v) for v in np.fromfile(f, count=3, dtype=np.int32))
            seen = int(np.fromfile(f, count=1, dtype=np.int32)[0])
            header = (major, minor, revision, seen)
        body = np.fromfile(f, dtype=np.float32)
    return header, body


def write_darknet_weights(weights_path, body, version):
    """Inverse of read_darknet_weights (used for the synthetic weight files)."""
    with open(weights_path, "wb") as f:
        if version == "v3":
            np.array([0, 2, 0, 0, 0], dtype=np.int32).tofile(f)
        else:
            np.array([0, 1, 0, 0], dtype=np.int32).tofile(f)
        np.ascontiguousarray(body, dtype=np.float32).tofile(f)


# ---- images ---------------------------------------------------------------------------------------
def load_image_paths(path_to_img_dir):
    """Files of the directory with an image extension, os.listdir order (reference net/base.py:64-66)."""
    root = os.path.abspath(path_to_img_dir)
    return [os.path.join(root, f) for f in os.listdir(path_to_img_dir) if f.lower().endswith(IMAGE_EXTENSIONS)]


def preprocess_image(image_path, new_shape):
    """decode -> stretch-resize to (h, w) -> RGB -> [0,1] float (reference net/base.py:115-155,
    which uses OpenCV: bilinear, no letterbox).  OpenCV is not available to this package; Pillow's
    bilinear resample stands in, so pixel values can differ from cv2.resize in the last bits
    (documented: parity for this step is unpinned)."""
    from PIL import Image
    try:
        img = Image.open(image_path)
        img.load()
    except Exception:
        print("Failed to read {}".format(image_path))
        return None
    img = img.convert("RGB").resize((int(new_shape[1]), int(new_shape[0])), Image.BILINEAR)
    return np.asarray(img, dtype=np.float64) / 255.


def decode_image(image_path):
    """JPEG/PNG decode to RGB uint8 [H,W,3] on the host (Pillow; the reference uses cv2.imread, net/base.py:117)."""
    from PIL import Image
    try:
        img = Image.open(image_path)
        img.load()
    except Exception:
        print("Failed to read {}".format(image_path))
        return None
    return np.array(img.convert("RGB"), dtype=np.uint8)         # a writable, contiguous copy


def preprocess_image_gpu(image_path, new_shape, device=None, stream=None):
    """preprocess_image with the resize / colour order / /255 on the device (yolo_preprocess_resize): OpenCV's 8-bit
    INTER_LINEAR arithmetic (what the reference's cv2.resize computes, restated in oracle/preprocess_ref.py) instead of
    Pillow's resampler.  Returns a float32 CUDA tensor [h, w, 3] or None when the file cannot be read."""
    import torch
    rgb = decode_image(image_path)
    if rgb is None:
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    src = torch.from_numpy(rgb).to(dev)
    dst = torch.empty((int(new_shape[0]), int(new_shape[1]), 3), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    _hip.check(_hip.lib().yolo_preprocess_resize(src.data_ptr(), rgb.shape[0], rgb.shape[1], rgb.shape[1] * 3, dst.data_ptr(),
                                                 int(new_shape[0]), int(new_shape[1]), 0, st), "yolo_preprocess_resize")
    return dst


def generate_test_batch_gpu(img_paths, batch_size, input_shape):
    """generate_test_batch with device-side preprocessing: yields ([B,h,w,3] float32 CUDA tensor, paths)."""
    import torch
    for start in range(0, len(img_paths), batch_size):
        chunk = img_paths[start:start + batch_size]
        images = []
        for p in chunk:
            image = preprocess_image_gpu(p, input_shape)
            if image is None:
                raise IOError("cannot read image {}".format(p))
            images.append(image)
        yield torch.stack(images, dim=0), chunk


def generate_test_batch(img_paths, batch_size, input_shape):
    """Yields ([B,h,w,c] float array, paths); the last batch may be short (reference net/base.py:158-168)."""
    for start in range(0, len(img_paths), batch_size):
        chunk = img_paths[start:start + batch_size]
        images = []
        for p in chunk:
            image = preprocess_image(p, input_shape)
            if image is None:
                raise IOError("cannot read image {}".format(p))
            images.append(image)
        yield np.stack(images, axis=0), chunk


_LABEL_FONT = None


def _label_font():
    """A scalable default font whose cap height is close to cv2.FONT_HERSHEY_SIMPLEX at scale 0.5 (~11 px)."""
    global _LABEL_FONT
    if _LABEL_FONT is None:
        from PIL import ImageFont
        try:
            _LABEL_FONT = ImageFont.load_default(12)
        except TypeError:               # Pillow without the sized default font
            _LABEL_FONT = ImageFont.load_default()
    return _LABEL_FONT


def draw_boxes(path_to_img, boxes, class_names, rgb=None):
    """Rectangles + "name prob" labels as the reference draws them (net/base.py:212-226): corners scaled by the ORIGINAL image
    size and clamped at 0 only, colour COLORS[class_idx % 6], `cv2.rectangle(..., thickness=3)` = a 3-pixel outline CENTRED on
    the corner coordinates (one pixel either side), label with its BASELINE's left end at (tl.x, tl.y - 10) -- so a box at the
    top edge loses its label, as there.  Pillow rasteriser (OpenCV is absent: glyph shapes differ, geometry does not).
    rgb: the image already decoded (uint8 [H,W,3], what decode_image returns) -- the pipelined test loop does not read and
    decode every file twice; same pixels.  Returns a PIL image."""
    from PIL import Image, ImageDraw
    image = Image.open(path_to_img).convert("RGB") if rgb is None else Image.fromarray(rgb, "RGB")
    w, h = image.size
    draw = ImageDraw.Draw(image)
    font = _label_font()
    for box in boxes:
        tl = np.maximum(box.get_top_left(h, w), 0)
        br = np.maximum(box.get_bottom_right(h, w), 0)
        tl, br = (int(tl[0]), int(tl[1])), (int(br[0]), int(br[1]))
        bgr = COLORS[box.class_idx % len(COLORS)]
        rgb = (bgr[2], bgr[1], bgr[0])
        x0, x1 = min(tl[0], br[0]), max(tl[0], br[0])
        y0, y1 = min(tl[1], br[1]), max(tl[1], br[1])
        draw.rectangle([(x0 - 1, y0 - 1), (x1 + 1, y1 + 1)], outline=rgb, width=3)
        text = "{} {:.3f}".format(class_names[box.class_idx], box.prob)
        try:
            draw.text((tl[0], tl[1] - 10), text, fill=rgb, font=font, anchor="ls")
        except (ValueError, TypeError):         # bitmap default font: no anchors; place its ~11-pixel box above the baseline
            draw.text((tl[0], tl[1] - 21), text, fill=rgb, font=font)
    return image


def save_image(image, out_path):
    out_dir = os.path.dirname(out_path)
    if out_dir and not os.path.isdir(out_dir):
        os.makedirs(out_dir, exist_ok=True)
    if out_path.lower().endswith(".png"):       # cv2.imwrite's default PNG setting (IMWRITE_PNG_COMPRESSION = 1), not Pillow's 6: same
        image.save(out_path, compress_level=1)  # pixels, a fifth of the encode time
    else:
        image.save(out_path)


def draw_save_task(path, rgb, records, class_names, out_dir):
    """One image of the test loop's last stage (reference net/yolo.py:88-95) as a task for a worker PROCESS of the pipelined loop
    (drawing holds the GIL: threads do not scale): records [(x, y, w, h, class_idx, prob)] -> boxes drawn on the decoded pixels ->
    `<stem>_out<ext>` written.  Touches neither torch nor the GPU.  Returns the console line."""
    boxes = [BoundingBox(r[0], r[1], r[2], r[3], 0, 0, r[4], r[5]) for r in records]
    new_img = draw_boxes(path, boxes, class_names, rgb=rgb)
    file_name, file_ext = os.path.splitext(os.path.basename(path))
    out_path = os.path.join(out_dir, "{}_out{}".format(file_name, file_ext))
    save_image(new_img, out_path)
    return "{}: Found {} objects. Saved to {}".format(file_name, len(boxes), out_path)
