"""Builds libppsfm_hip.so (gfx950) in-tree with hipcc.  No CPU fallback exists: if the library is
missing or stale and hipcc is unavailable, importing the bindings fails loudly."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libppsfm_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(os.path.dirname(_HERE), "include", "ppsfm_hip.h")]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_library(force=False, verbose=False, jobs=8):
    """Compile every .hip translation unit for gfx950 and link the shared library."""
    if not force and not is_stale():
        return LIB
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
    procs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [src] + _deps()[len(sources()):]):
            continue
        cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        while len(procs) >= jobs:
            _wait(procs.pop(0))
    for p in procs:
        _wait(p)
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def _wait(item):
    src, proc = item
    out, _ = proc.communicate()
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    text = out.decode(errors="replace").strip()
    if text:
        print(text)


HOST_SAN_LIB = os.path.join(_HERE, "build", "host_san", "libppsfm_host_san.so")


def build_host_sanitized(force=False, verbose=False, jobs=8):
    """Every translation unit with its HOST half under -fsanitize=address,undefined (-fno-gpu-sanitize: the device code is compiled as usual - a
    host-only build leaves the launch stubs' code objects undefined), linked into build/host_san/libppsfm_host_san.so - the image ordering, the Cholesky task planner, the host
    pair-list builder, the sampler and the exception containment run under the sanitizers without a device (tests/test_host_sanitizers.py drives
    them through the C ABI from tests/host_sanitizer_driver.cpp).  Not the product library: nothing else loads it."""
    objdir = os.path.dirname(HOST_SAN_LIB)
    os.makedirs(objdir, exist_ok=True)
    deps = _deps()
    if not force and os.path.exists(HOST_SAN_LIB) and all(os.path.getmtime(p) <= os.path.getmtime(HOST_SAN_LIB) for p in deps):
        return HOST_SAN_LIB
    flags = ["-O1", "-g", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
             "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-Wno-unused-function", "-w"]
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        while len(procs) >= jobs:
            _wait(procs.pop(0))
    for p in procs:
        _wait(p)
    subprocess.check_call([hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan", "-o", HOST_SAN_LIB] + objs)
    return HOST_SAN_LIB


def sanitizer_runtime_dir():
    """where libclang_rt.asan-x86_64.so lives (the driver's rpath)"""
    out = subprocess.check_output([hipcc(), "--print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    if not os.path.isabs(out):
        out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "--print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    return os.path.dirname(out)


if __name__ == "__main__":
    import sys
    if "--host-asan" in sys.argv:
        print(build_host_sanitized(force="--force" in sys.argv, verbose=True))
    else:
        build_library(force="--force" in sys.argv, verbose=True)
        print(LIB)
