#!/usr/bin/env python3
"""Oracle (test infrastructure): BASELINE.json config 1's literal input as a fixture.

Reads the reference's img/dog.jpg (768x576 JPEG) in the build container, decodes it (Pillow; the reference uses
cv2.imread, net/base.py:117 -- OpenCV is absent, so the decoder is unpinned), stretches it to the 416x416 network input
with the restated OpenCV 8-bit INTER_LINEAR arithmetic (oracle/preprocess_ref.py; cv2.resize, net/base.py:121) and
stores the RGB uint8 result -- DATA (pixels), not source -- as tests/golden/dog_416_rgb_u8.npz.  The reference's
x_batch for this image is that array / 255. (net/base.py:153).  /root/reference never travels to the GPU box.

    python oracle/gen_dog.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("YOLO_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import preprocess_ref  # noqa: E402


def main():
    src = os.path.join(REF, "img", "dog.jpg")
    rgb = np.asarray(Image.open(src).convert("RGB"), dtype=np.uint8)
    small = preprocess_ref.resize_linear_u8(rgb, 416, 416)
    out = os.path.join(ROOT, "tests", "golden", "dog_416_rgb_u8.npz")
    np.savez_compressed(out, rgb=small, source=np.array("reference img/dog.jpg %dx%d, Pillow %s decode, oracle/preprocess_ref.py resize"
                                                        % (rgb.shape[1], rgb.shape[0], Image.__version__)))
    print(out, small.shape, small.dtype, os.path.getsize(out), "bytes; source", rgb.shape)


if __name__ == "__main__":
    main()
