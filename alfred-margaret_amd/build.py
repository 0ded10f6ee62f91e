"""Builds the native libraries in-tree (alfred-margaret_amd/lib/*.so) for gfx950.

  libam.so           the product: C ABI (include/am.h) + HIP kernels, hipcc --offload-arch=gfx950
  libam_host.so      C++ host mirror of the reference API (host/), links libam.so
  libam_imgcheck.so  TEST-ONLY host interpreter of the device image (g++, no HIP)
  libam_synth.so     synthetic haystack generator (bench/test input; device kernel + identical host loop)

hipcc cross-compiles without a GPU, so this runs in the build container; the .so files travel to
the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "lib")
ARCH = "gfx950"
LINK_EXTRA = ["-ldl"]        # RCCL (am_multi_*) is bound with dlopen at first use, not at link time


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _glob_deps(*dirs):
    out = []
    for d in dirs:
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp", ".inc", ".cpp", ".hip")):
                out.append(os.path.join(d, f))
    return out


def build_libam(force=False):
    """Every source is compiled to its own object (in parallel: am_kernels.hip alone takes ~40 s), then linked."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB, exist_ok=True)
    obj_dir = os.path.join(LIB, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    target = os.path.join(LIB, "libam.so")
    names = ("am_abi.cpp", "am_replacer.cpp", "am_contains_all.cpp", "am_flatten.cpp", "am_kernels.hip", "am_sfx.hip", "am_scan.hip", "am_replace.hip", "am_rploop.hip", "am_dense.hip", "am_multi.cpp")
    srcs = [os.path.join(CSRC, f) for f in names if os.path.exists(os.path.join(CSRC, f))]
    headers = [d for d in _glob_deps(CSRC) if d.endswith((".h", ".hpp", ".inc"))] + [os.path.join(ROOT, "include", "am.h")]
    objs = [os.path.join(obj_dir, os.path.basename(f) + ".o") for f in srcs]

    def compile_one(pair):
        src, obj = pair
        if force or _stale(obj, [src] + headers):
            subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o", obj + ".tmp"])
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(len(srcs)) as pool:
        list(pool.map(compile_one, zip(srcs, objs)))
    if force or _stale(target, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", *objs, "-o", target + ".tmp"] + LINK_EXTRA
        subprocess.check_call(cmd)
        os.replace(target + ".tmp", target)
    return target


def build_imgcheck(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libam_imgcheck.so")
    srcs = [os.path.join(CSRC, f) for f in ("am_imgcheck.cpp", "am_flatten.cpp")]
    if force or _stale(target, _glob_deps(CSRC)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", *srcs, "-o", target + ".tmp"])
        os.replace(target + ".tmp", target)
    return target


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libam_host.so")
    libam = build_libam(force)
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".cpp")]
    deps = _glob_deps(HOST) + [libam, os.path.join(ROOT, "include", "am.h")]
    if force or _stale(target, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), *srcs,
               "-L", LIB, "-lam", "-Wl,-rpath,$ORIGIN", "-o", target + ".tmp"]
        subprocess.check_call(cmd)
        os.replace(target + ".tmp", target)
    return target


def build_synth(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libam_synth.so")
    srcs = [os.path.join(CSRC, "am_synth.hip")]
    if force or _stale(target, srcs + [os.path.join(CSRC, "am_synth.h")]):
        subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", *srcs, "-o", target + ".tmp"])
        os.replace(target + ".tmp", target)
    return target


def build_all(force=False):
    return {"libam": build_libam(force), "libam_host": build_host(force), "libam_imgcheck": build_imgcheck(force),
            "libam_synth": build_synth(force)}


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv))
