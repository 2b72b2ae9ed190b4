"""TEST INFRASTRUCTURE: build and bind the host-emulation build of the kernel sources
(tests/emu/hip/hip_runtime.h).  Used by CPU tests to check kernel logic; never by the product."""
import ctypes as C
import glob
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "audio-denoising_amd", "csrc")
ASAN = os.environ.get("DN_EMU_ASAN") == "1"      # AddressSanitizer build of the host emulation (run pytest under LD_PRELOAD=libasan)
OUT = os.path.join(HERE, "build", "libdn_emu_asan.so" if ASAN else "libdn_emu.so")
sys.path.insert(0, REPO)


def build(force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "dn_cpx.hpp"),
                                                           os.path.join(REPO, "include", "dn_denoise.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    opt = ["-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer"] if ASAN else ["-O2"]
    cmd = ["g++", "-std=c++17"] + opt + ["-fPIC", "-shared", "-pthread", "-x", "c++", "-I", HERE] + srcs + ["-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


def load():
    from audio_denoising_amd._lib import DnLib
    return DnLib(build())


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
