"""Builds the native libraries in-tree (alfred-margaret_amd/lib/*.so) for gfx950.

  libam.so           the product: C ABI (include/am.h) + HIP kernels, hipcc --offload-arch=gfx950, -fvisibility=hidden (exports = am.h + am_debug.h)
  libam_host.so      C++ host mirror of the reference API (host/), links libam.so
  libam_check.so     TEST-ONLY (tests/native/am_ac.hip): k_ac, the general AC-walk kernel = the parity gate's independent second algorithm;
                     registers itself with libam when loaded (tests, bench.py's gate, smoke()); the product never loads it
  libam_imgcheck.so  TEST-ONLY (tests/native/am_imgcheck.cpp) host interpreter of the device image (g++, no HIP)
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
NATIVE_TESTS = os.path.join(ROOT, "tests", "native")
# AM_LIB_DIR: build into / load from another directory (tools/sanitize.sh keeps its instrumented libraries apart from the product's)
LIB = os.environ.get("AM_LIB_DIR") or os.path.join(PKG, "lib")
ARCH = "gfx950"
LINK_EXTRA = ["-ldl"]        # RCCL (am_multi_*) is bound with dlopen at first use, not at link time
# AM_SANITIZE=1 (tools/sanitize.sh): every HOST translation unit -- the C ABI's runtime, the flattener, the Replacer's and multi-GPU host code, the host mirror, the
# image interpreter -- with AddressSanitizer + UndefinedBehaviorSanitizer (clang's, shared runtime: python is not instrumented and gets it through LD_PRELOAD).  The
# kernels are compiled as always: GPU sanitizers are not available on this pool.
SANITIZE = bool(os.environ.get("AM_SANITIZE"))
# AM_BOUNDS_CHECK=1 (tools/bounds_check.sh): the kernels compiled with their index assertions (csrc/am_bounds.h); run on the GPU box over the parity tests
BOUNDS = ["-DAM_BOUNDS_CHECK"] if os.environ.get("AM_BOUNDS_CHECK") else []
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined", "-shared-libsan", "-fno-omit-frame-pointer", "-g", "-O1"] if SANITIZE else []
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


def _host_cxx():
    """g++ for the host-only libraries; clang++ under AM_SANITIZE (one sanitizer runtime for every library of the process)."""
    return CLANGXX if SANITIZE else "g++"


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
    names = ("am_abi.cpp", "am_replacer.cpp", "am_contains_all.cpp", "am_flatten.cpp", "am_kernels.hip", "am_dfa.hip", "am_scan.hip", "am_replace.hip", "am_rploop.hip", "am_rplds.hip", "am_dense.hip", "am_multi.cpp")
    srcs = [os.path.join(CSRC, f) for f in names if os.path.exists(os.path.join(CSRC, f))]
    headers = [d for d in _glob_deps(CSRC) if d.endswith((".h", ".hpp", ".inc"))] + [os.path.join(ROOT, "include", "am.h"), os.path.join(ROOT, "include", "am_debug.h")]
    objs = [os.path.join(obj_dir, os.path.basename(f) + ".o") for f in srcs]

    def compile_one(pair):
        src, obj = pair
        if force or _stale(obj, [src] + headers):
            san = SAN_FLAGS if src.endswith(".cpp") else []          # host translation units only
            subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", *san, *BOUNDS, "-x", "hip", "-c", src, "-o", obj + ".tmp"])
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(len(srcs)) as pool:
        list(pool.map(compile_one, zip(srcs, objs)))
    if force or _stale(target, objs + [os.path.join(CSRC, "libam.map")]):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", *objs, "-Wl,--version-script=" + os.path.join(CSRC, "libam.map"), "-o", target + ".tmp"] + LINK_EXTRA + \
              (["-fsanitize=address,undefined", "-shared-libsan"] if SANITIZE else [])
        subprocess.check_call(cmd)
        os.replace(target + ".tmp", target)
    return target


def build_imgcheck(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libam_imgcheck.so")
    srcs = [os.path.join(NATIVE_TESTS, "am_imgcheck.cpp"), os.path.join(CSRC, "am_flatten.cpp")]
    if force or _stale(target, _glob_deps(CSRC) + srcs):
        subprocess.check_call([_host_cxx(), "-O2", "-std=c++17", "-fPIC", "-shared", *SAN_FLAGS, "-I", CSRC, *srcs, "-o", target + ".tmp"])
        os.replace(target + ".tmp", target)
    return target


def build_check(force=False):
    """libam_check.so: k_ac (tests/native/am_ac.hip) against libam's own headers; links libam.so for the registration call."""
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libam_check.so")
    libam = build_libam(force)
    src = os.path.join(NATIVE_TESTS, "am_ac.hip")
    deps = [src, libam] + [d for d in _glob_deps(CSRC) if d.endswith((".h", ".inc"))] + [os.path.join(ROOT, "include", "am_debug.h")]
    if force or _stale(target, deps):
        subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-shared", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                               src, "-L", LIB, "-lam", "-Wl,-rpath,$ORIGIN", "-o", target + ".tmp"])
        os.replace(target + ".tmp", target)
    return target


def build_rccl_stub(force=False):
    """TEST-ONLY libam_rccl_stub.so (tests/native/rccl_stub.cpp), soname librccl.so.1: file-based stand-ins for the RCCL calls of csrc/am_multi.cpp, so
    that its N-rank code runs as N processes on a one-GPU box (tests/test_multi_ranks.py).  Not part of build_all()'s product set, built on demand."""
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libam_rccl_stub.so")
    src = os.path.join(NATIVE_TESTS, "rccl_stub.cpp")
    if force or _stale(target, [src]):
        subprocess.check_call([_hipcc(), "-O1", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-shared", "-x", "hip", "--offload-arch=" + ARCH, src,
                               "-Wl,-soname,librccl.so.1", "-o", target + ".tmp"])
        os.replace(target + ".tmp", target)
    return target


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libam_host.so")
    libam = build_libam(force)
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".cpp")]
    deps = _glob_deps(HOST) + [libam, os.path.join(ROOT, "include", "am.h")]
    if force or _stale(target, deps):
        cmd = [_host_cxx(), "-O2", "-std=c++17", "-fPIC", "-shared", *SAN_FLAGS, "-I", os.path.join(ROOT, "include"), *srcs,
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
    return {"libam": build_libam(force), "libam_host": build_host(force), "libam_check": build_check(force), "libam_imgcheck": build_imgcheck(force),
            "libam_synth": build_synth(force), "libam_rccl_stub": build_rccl_stub(force)}


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv))
