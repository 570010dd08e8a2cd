"""Build libctc_amd.so (the C-ABI library, include/ctc_amd.h) with hipcc for gfx950.

    python -m ctc_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is built IN-TREE (ctc_amd/lib/), is
git-ignored, and travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libctc_amd.so")
SOURCES = ["api.hip", "noblank.hip", "binary.hip", "blank.hip", "decode.hip", "targets.hip", "producer.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libctc_amd.so")
    return exe


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(HERE, "..", "include", "ctc_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), name=None):
    """name=None: the product library.  name="diag" (python -m ctc_amd.build --diag): the same sources with
    -DCTC_AMD_DIAGNOSTICS -> lib/libctc_amd_diag.so, the library tools/ load (phase stamps, forced kernel
    choices, the chain probe); any other name/flags: a variant for A/B runs or fault injection."""
    so = SO if name is None else os.path.join(LIBDIR, "libctc_amd_%s.so" % name)
    if name is None and any(f.startswith(("-DCTC_X_", "-DCTC_AMD_EXPERIMENTS", "-DCTC_AMD_FAULT_INJECT",
                                          "-DCTC_AMD_DIAGNOSTICS")) for f in extra_flags):
        raise RuntimeError("ctc_amd.build: experiment / diagnostics / fault-injection flags %s are refused for the "
                           "product library -- give the variant a name" % (list(extra_flags),))
    if name is None and not force and not _stale():
        return so
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" if name is None else "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so + ".tmp", *objs])
    os.replace(so + ".tmp", so)
    return so


def build_diag(verbose=False):
    return build(force=True, verbose=verbose, extra_flags=("-DCTC_AMD_DIAGNOSTICS",), name="diag")


def build_fault(verbose=False):
    """The fault-injection variant tests/test_status.py drives (-DCTC_AMD_FAULT_INJECT -> lib/libctc_amd_fault.so):
    built here, with the product library, so that the GPU run compiles nothing."""
    so = os.path.join(LIBDIR, "libctc_amd_fault.so")
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(build()):
        return so
    return build(force=True, verbose=verbose, extra_flags=("-DCTC_AMD_FAULT_INJECT",), name="fault")


def build_occupant():
    """tools/micro/coresident.hip -> lib/libcoresident.so: the stand-in for RCCL's collective kernel that the co-residence
    measurement (tools/coresident.py, tests/test_bench_gpu.py, `bench.py --rehearse-collective`) launches beside the loss
    kernel.  Diagnostics only -- not linked into and not loaded by the product library."""
    src = os.path.join(HERE, "..", "tools", "micro", "coresident.hip")
    so = os.path.join(LIBDIR, "libcoresident.so")
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    os.makedirs(LIBDIR, exist_ok=True)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so + ".tmp"])
    os.replace(so + ".tmp", so)
    return so


HOST_EXT_DIR = os.path.join(LIBDIR, "ext")
HOST_EXT_SO = os.path.join(HOST_EXT_DIR, "ctc_amd_autograd_ext.so")


HOST_EXT_STAMP = os.path.join(HOST_EXT_DIR, "built_for_torch.txt")


def host_ext_is_current():
    """the extension exists, is newer than its source and was built against the torch that is imported now"""
    import torch
    src = os.path.join(CSRC, "autograd_ext.cpp")
    if not (os.path.exists(HOST_EXT_SO) and os.path.exists(HOST_EXT_STAMP)):
        return False
    if os.path.getmtime(HOST_EXT_SO) < os.path.getmtime(src):
        return False
    return open(HOST_EXT_STAMP).read().strip() == torch.__version__


def build_host_ext(verbose=False):
    """csrc/autograd_ext.cpp -> lib/ext/ctc_amd_autograd_ext.so: the C++ autograd node around the C-ABI calls (host code
    only, g++ through torch.utils.cpp_extension, in-tree so that it travels with the snapshot).  Optional: without it
    the Python Functions of ctc_amd/functional.py issue the same launches."""
    import torch
    src = os.path.join(CSRC, "autograd_ext.cpp")
    if host_ext_is_current():
        return HOST_EXT_SO
    from torch.utils import cpp_extension
    os.makedirs(HOST_EXT_DIR, exist_ok=True)
    cpp_extension.load(name="ctc_amd_autograd_ext", sources=[src], build_directory=HOST_EXT_DIR, extra_cflags=["-O2"],
                       with_cuda=False, verbose=verbose, is_python_module=False)
    with open(HOST_EXT_STAMP, "w") as f:
        f.write(torch.__version__)
    return HOST_EXT_SO


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
