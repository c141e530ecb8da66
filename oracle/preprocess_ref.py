"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's image preprocessing (net/base.py:115-155):
cv2.imread -> cv2.resize(image, (w, h)) [INTER_LINEAR, stretch, no letterbox] -> BGR->RGB -> / 255.

PARITY UNPINNED: the arithmetic lives in OpenCV (requirements.txt: opencv-python, not installable here; the reference
holds no resized fixtures).  This restates OpenCV's published 8-bit INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp:
half-pixel centres, coefficients in 11-bit fixed point `saturate_cast<short>(c * 2048)`, horizontal pass in int32, vertical
pass `((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`), so that the HIP kernel has a bit-exact checker.
Only tests/ may import this module."""
import numpy as np


def _coeffs(src, dst):
    """per destination index: source index s0 (clamped), s1 and the two 11-bit integer weights"""
    scale = float(src) / float(dst)                                  # double, like OpenCV's scale_x
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)                 # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0; s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0; s[hi] = src - 1
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)      # cvRound: round half to even
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, w0, w1


def resize_linear_u8(img, dst_h, dst_w):
    """img: uint8 [H, W, C] -> uint8 [dst_h, dst_w, C], OpenCV INTER_LINEAR for 8-bit images (stretch)."""
    img = np.asarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    if (H, W) == (dst_h, dst_w):
        return img.copy()
    x0, x1, a0, a1 = _coeffs(W, dst_w)
    y0, y1, b0, b1 = _coeffs(H, dst_h)
    src = img.astype(np.int64)
    rows0 = src[y0][:, x0] * a0[None, :, None] + src[y0][:, x1] * a1[None, :, None]      # hresize, int32 range
    rows1 = src[y1][:, x0] * a0[None, :, None] + src[y1][:, x1] * a1[None, :, None]
    out = (((b0[:, None, None] * (rows0 >> 4)) >> 16) + ((b1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess(img_rgb_u8, new_shape):
    """RGB uint8 [H,W,3] -> float32 [h,w,3] in [0,1]: resize, then / 255. in float64, cast to float32 (the reference
    feeds the float64 array to a float32 placeholder, net/base.py:155 + net/yolo.py:83)."""
    r = resize_linear_u8(img_rgb_u8, int(new_shape[0]), int(new_shape[1]))
    return (r.astype(np.float64) / 255.).astype(np.float32)
