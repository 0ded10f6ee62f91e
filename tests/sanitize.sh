#!/bin/bash
# Sanitiser pass over everything under the C ABI that runs on the host (SURVEY 5 "race detection / sanitizers").
#   1. AddressSanitizer + UndefinedBehaviorSanitizer builds (clang, shared runtime) of: the oracle (oracle/am_oracle.c), the flattener + the host interpreter of the
#      image (libam_imgcheck.so: am_flatten.cpp + the walk code of am_image.h that the kernels share), the host mirror (libam_host.so) and every host translation unit
#      of libam.so (am_abi.cpp, am_replacer.cpp, am_contains_all.cpp, am_flatten.cpp, am_multi.cpp) -- into build/sanitize/, apart from the product's libraries;
#   2. the whole CPU suite (`-m "not gpu"`) on them, python getting the runtime through LD_PRELOAD; any report aborts the run.
# The kernels are not instrumented (GPU sanitizers are not available on this pool); their index arithmetic is what the -DAM_BOUNDS_CHECK build of the
# kernels checks on the GPU box (tools/bounds_check.sh).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)      # (lives under tests/: it builds and runs the oracle, which only test infrastructure may)
OUT=$R/build/sanitize
mkdir -p "$OUT"
CLANG=/opt/rocm/lib/llvm/bin/clang
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
$CLANG -O1 -g -std=c11 -Wall -Wextra -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -fno-omit-frame-pointer -shared -o "$OUT/libam_oracle.so" "$R/oracle/am_oracle.c"
export AM_LIB_DIR=$OUT AM_SANITIZE=1 AM_ORACLE_LIB=$OUT/libam_oracle.so
cd "$R"
python -c "import alfred_margaret_amd as am; print(sorted(am.build.build_all()))"
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
# (the plain-C consumers of tests/c are linked by gcc against libam.so: with the instrumented library they would need the sanitiser runtime on their link line -- left out)
python -m pytest tests -x -q -m "not gpu" -k "not c_driver and not multi_driver" "$@"
echo "sanitize.sh: no report"
