"""torch operator registration of the render path: ``torch.ops.diner.render`` / ``torch.ops.diner.render_image``
(diner_amd/csrc/torch_ops.cpp -- dispatcher ops that call the C ABI of libdiner_hip.so on the caller's current stream).

north_star asks for the plug-in "via a torch C++/HIP extension"; the library itself is bound with ctypes (``_lib.py``), and this
small host-only extension puts the two whole-path entry points behind the torch dispatcher as well.  ``NeRFRendererDGS`` uses
the ops when the extension is built (``binding = "torch_ops"``), the ctypes binding otherwise and for the build-only
arguments (replayed noise, per-stage events); both call the same C functions.
"""
from __future__ import annotations

import subprocess
from pathlib import Path

import torch

from . import _lib

SRC = Path(__file__).resolve().parent / "csrc" / "torch_ops.cpp"
LIB = _lib.LIB_PATH.parent / "libdiner_torch_ops.so"
_loaded = False


def build(force: bool = False) -> Path:
    """g++ the extension against the installed torch headers (host code only: seconds); needs libdiner_hip.so next to it."""
    if LIB.exists() and not force and LIB.stat().st_mtime >= max(SRC.stat().st_mtime, _lib.LIB_PATH.stat().st_mtime):
        return LIB
    from torch.utils import cpp_extension as ce
    inc = [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include"]
    libs = [f"-L{p}" for p in ce.library_paths()]
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, str(SRC), "-o", str(LIB), *libs,
           "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", f"-L{LIB.parent}", "-ldiner_hip", "-Wl,-rpath,$ORIGIN"]
    subprocess.run(cmd, check=True)
    return LIB


def available() -> bool:
    return LIB.exists()


def load():
    """Register the ops (idempotent); raises if the extension has not been built."""
    global _loaded
    if not _loaded:
        if not LIB.exists():
            raise RuntimeError(f"{LIB} is missing: run `python __graft_entry__.py` (build()) first")
        _lib.lib()                       # libdiner_hip.so first: the extension resolves its symbols from it
        torch.ops.load_library(str(LIB))
        _loaded = True
    return torch.ops.diner
