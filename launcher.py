#!/usr/bin/env python3
"""`python launcher.py --config <ini> --mode test` -- same command line as the reference's launcher.py,
served by the HIP backend (tensorflow-yolo_amd/launcher.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tensorflow_yolo_amd.launcher import main  # noqa: E402

if __name__ == "__main__":
    main()
