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
    if not force and not stale():
        return LIB
    from torch.utils import cpp_extension as ce
    inc = [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include"]
    libs = [f"-L{p}" for p in ce.library_paths()]
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, str(SRC), "-o", str(LIB), *libs,
           "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", f"-L{LIB.parent}", "-ldiner_hip", "-Wl,-rpath,$ORIGIN"]
    STAMP.unlink(missing_ok=True)
    subprocess.run(cmd, check=True)
    STAMP.write_text(_deps_digest() + "\n")
    return LIB


STAMP = LIB.with_suffix(".stamp")


_digest_cache = {}


def _deps_digest() -> str:
    """sha256 over what the extension is built from and links against: torch_ops.cpp, the ABI header, libdiner_hip.so itself
    (contents, not mtimes: the snapshot that carries the build to the GPU box need not preserve times).  Hashed once per process and
    state of the files (every renderer constructor asks)."""
    import hashlib
    deps = (SRC, SRC.parents[2] / "include" / "diner_hip.h", _lib.LIB_PATH)
    key = tuple((str(p), p.stat().st_mtime_ns, p.stat().st_size) if p.exists() else (str(p), 0, 0) for p in deps)
    if key not in _digest_cache:
        h = hashlib.sha256()
        for p in deps:
            h.update(p.read_bytes() if p.exists() else b"<missing>")
        _digest_cache.clear()
        _digest_cache[key] = h.hexdigest()
    return _digest_cache[key]


def stale() -> bool:
    """True when the extension is missing or was built from / against other files than the ones present (libdiner_hip.so,
    torch_ops.cpp, the ABI header): such a file may carry another argument list and must not be loaded."""
    return not (LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == _deps_digest())


def available() -> bool:
    """The renderer takes the torch-ops binding only when the extension is built AND current; otherwise ctypes."""
    return not stale()


def load():
    """Register the ops (idempotent); raises if the extension has not been built."""
    global _loaded
    if not _loaded:
        if not LIB.exists():
            raise RuntimeError(f"{LIB} is missing: run `python __graft_entry__.py` (build()) first")
        if stale():
            raise RuntimeError(f"{LIB} is older than libdiner_hip.so / torch_ops.cpp / diner_hip.h: rebuild it (diner_amd.ops.build())")
        hip = _lib.lib()                 # libdiner_hip.so first: the extension resolves its symbols from it
        torch.ops.load_library(str(LIB))
        built_for = int(torch.ops.diner.abi_version())
        if built_for != hip.diner_version():   # (the ops check this again at every entry)
            raise RuntimeError(f"{LIB} was built for ABI version {built_for}, libdiner_hip.so is {hip.diner_version()}: "
                               "rebuild it (diner_amd.ops.build(force=True))")
        _loaded = True
    return torch.ops.diner
