"""CPU: the host code of the planner and the C ABI (plan.cpp, api.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer
(VERDICT r4 #4 ii).  `make SAN=1` builds a second library (host side instrumented, device code as usual: GPU ASan is not
available on this pool and is not asked for); the planning / ABI tests of tests/test_abi.py and tests/test_host_logic.py and the
weight-packing probe run against it in child processes with the ASan runtime preloaded.  Any report aborts the child.
(First run of this build, round 5: `net->dev_weights + offset` on a plan without weights -- an offset applied to a null pointer in
yolo_net_kernel_info; fixed in api.cpp: weights_at.)"""
import glob
import os
import subprocess
import sys

import pytest

from helpers import ROOT

CSRC = os.path.join(ROOT, "tensorflow-yolo_amd", "csrc")
SAN_DIR = os.path.join(CSRC, "build_san")
SAN_LIB = os.path.join(SAN_DIR, "libyolo_hip_san.so")


def _asan_runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


@pytest.fixture(scope="module")
def san_env():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no libclang_rt.asan-x86_64.so under /opt/rocm")
    subprocess.check_call(["make", "-C", CSRC, "-j4", "SAN=1", "OBJDIR=" + SAN_DIR, "OUT=" + SAN_LIB], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=rt, YOLO_HIP_LIB=SAN_LIB,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return env


def _run(cmd, env):
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    return r.stdout


def test_planning_and_abi_tests_pass_under_asan_ubsan(san_env):
    out = _run([sys.executable, "-m", "pytest", "tests/test_abi.py", "tests/test_host_logic.py", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                "-k", "not register_allocation and not do_not_spill"], san_env)
    assert " passed" in out and "failed" not in out, out[-2000:]


def test_weight_packing_under_asan_ubsan(san_env):
    out = _run([sys.executable, os.path.join("tests", "san_pack_probe.py")], san_env)
    assert "pack probe OK" in out, out[-2000:]
