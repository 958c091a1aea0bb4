"""Build libimagen_hip.so (gfx950) in-tree with hipcc.

The library is linked against the HIP runtime that ships inside the torch wheel
(torch/lib/libamdhip64.so, soname libamdhip64.so.7) and rpath'd to it first, so the
process holds exactly one HIP runtime and torch's device pointers / streams are
valid inside our kernels.  /opt/rocm/lib is the fallback rpath
for use without torch.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["igemm.hip", "conv_dma.hip", "conv_stream.hip", "conv_pw.hip", "conv_big.hip", "conv_pro.hip", "conv_gemm.hip", "conv_small.hip", "rowchain.hip", "elementwise.hip", "attention.hip", "sampler.hip", "temporal.hip", "codesize.hip", "probe.hip", "capi.hip"]
LIB = os.path.join(HERE, "libimagen_hip.so")
STAMP = os.path.join(HERE, ".libimagen_hip.stamp")


def _torch_lib_dir():
    try:
        import torch  # noqa: WPS433
        return os.path.join(os.path.dirname(torch.__file__), "lib")
    except Exception:  # pragma: no cover
        return None


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    with open(os.path.join(ROOT, "include", "imagen_hip.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == digest:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libimagen_hip.so")
    objs = []
    build_dir = os.path.join(HERE, "build")
    os.makedirs(build_dir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    procs = []
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append((src, subprocess.Popen(common + ["-c", os.path.join(CSRC, src), "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    tl = _torch_lib_dir()
    if tl and os.path.exists(os.path.join(tl, "libamdhip64.so")):
        link += ["-L" + tl, "-Wl,-rpath," + tl]
    link += ["-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout.decode())
    with open(STAMP, "w") as f:
        f.write(digest)
    if verbose:
        print(f"built {LIB}")
    return LIB


def build_variant(tag: str, defs, sources=None) -> str:
    """A/B library libimagen_hip_<tag>.so: the product sources with extra -D flags (bench-only builds, e.g. conv_big's timing ablations)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    build_dir = os.path.join(HERE, "build", tag)
    os.makedirs(build_dir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + list(defs)
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append((src, subprocess.Popen(common + ["-c", os.path.join(CSRC, src), "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    lib = os.path.join(HERE, f"libimagen_hip_{tag}.so")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    tl = _torch_lib_dir()
    if tl and os.path.exists(os.path.join(tl, "libamdhip64.so")):
        link += ["-L" + tl, "-Wl,-rpath," + tl]
    link += ["-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout.decode())
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
