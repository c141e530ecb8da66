"""Command-line entry of the HIP backend.

Keeps the reference launcher's surface (reference launcher.py:15-60): the two flags `--config` and
`--mode` with the same defaults (mode defaults to "anchor", exactly as there), `.ini` sections
COMMON / ANCHOR / TRAIN / TEST merged as {**section, **COMMON}, relative `*_dir` / `*_path` values
resolved against the .ini's directory, `anchors` / `class_names` parsed as Python literals, and the
network picked by COMMON.version.  Only `test` runs on this backend; `train` and `anchor` end with a
clear message.  Extra, optional keys: `dtype` (fp32 | fp16), `nms_mode` (agnostic | per_class), `max_boxes` / `cand_capacity` (record caps), `autotune` (True: per-layer tile timing at start-up);
version additionally accepts `v2-tiny`.  `--section` selects another TEST-like section (the
reference's yolo_2.ini keeps its COCO settings in [TEST_COCO], which no mode reaches there).
"""
import argparse
import ast
import configparser
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONFIG = os.path.join(_PKG, "config", "yolo_2.ini")
LITERAL_KEYS = ("anchors", "class_names")
PATH_SUFFIXES = ("_dir", "_path")


def resolve_section(values, ini_path):
    """One section's key/value strings -> dict with absolute paths and parsed literals."""
    base = os.path.dirname(os.path.abspath(ini_path))
    out = {}
    for key, value in values.items():
        if key.endswith(PATH_SUFFIXES) and not os.path.isabs(value):
            value = os.path.join(base, value)
        elif key in LITERAL_KEYS:
            value = ast.literal_eval(value)
        out[key] = value
    return out


def read_config(ini_path):
    parser = configparser.ConfigParser()
    if not parser.read(ini_path):
        raise IOError("cannot read config file {}".format(ini_path))
    return {name: resolve_section(dict(parser.items(name)), ini_path) for name in parser.sections()}


def pick_model(version):
    from .net.yolo import YoloV2, YoloV2Tiny, YoloV3
    table = {"v2": YoloV2, "v3": YoloV3, "v2-tiny": YoloV2Tiny}
    if version not in table:
        raise ValueError("Unsupported version: {}".format(version))
    return table[version]()


def run(cfg, mode, section=None):
    yolo = pick_model(cfg["COMMON"]["version"])
    if mode == "test":
        params = dict(cfg[section or "TEST"])
        params.update(cfg["COMMON"])
        yolo.test(params)
    elif mode in ("train", "anchor"):
        raise SystemExit("mode '{}' is not supported by the HIP inference backend (TEST mode only)".format(mode))
    else:
        raise ValueError("Unsupported mode: {}".format(mode))


def main(argv=None):
    ap = argparse.ArgumentParser(description="YOLO v2/v3 TEST-mode inference on MI355X")
    ap.add_argument("--config", dest="config", help="Path to configuration file", default=DEFAULT_CONFIG)
    ap.add_argument("--mode", dest="mode", help="Mode: (train|test|anchor)", default="anchor")
    ap.add_argument("--section", dest="section", help="section to use for test mode (default TEST)", default=None)
    args = ap.parse_args(argv)
    group = init_distributed()
    try:
        run(read_config(args.config), args.mode.lower(), args.section)
    finally:
        if group:
            import torch.distributed as dist
            dist.destroy_process_group()


def init_distributed():
    """One process per GPU: under a multi-process launcher (`torchrun launcher.py ...` sets WORLD_SIZE / RANK / LOCAL_RANK /
    MASTER_*) bind this process to its GPU and join the group BEFORE anything touches the device, so that Yolo.test shards
    every batch over the ranks (net/dist.py).  Backend "nccl" (= RCCL) on GPUs, "gloo" where there is none (the CPU tests).
    Returns True when this call created the group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return False
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() > local_rank and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return True


if __name__ == "__main__":
    if __package__ in (None, ""):       # executed as a script: make the package importable
        sys.path.insert(0, os.path.dirname(_PKG))
        import tensorflow_yolo_amd.launcher as _self
        _self.main()
    else:
        main()
