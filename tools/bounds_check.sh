#!/bin/bash
# Index assertions inside the kernels (csrc/am_bounds.h), run ON THE GPU BOX: GPU sanitizers are not available on this pool, so a -DAM_BOUNDS_CHECK build of libam
# checks its own LDS queue indices, pool slots and image offsets (k_sf, k_dfa / k_dfa_place, k_rp_lds) while the parity tests run over it; tests/conftest.py fails the
# session if am_debug_bounds_report counts anything.  The instrumented libraries live in build/bounds/ (AM_LIB_DIR), apart from the product's.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/bounds_check.sh'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export AM_LIB_DIR=$R/build/bounds AM_BOUNDS_CHECK=1
mkdir -p "$AM_LIB_DIR"
cd "$R"
python -c "import alfred_margaret_amd as am; print(sorted(am.build.build_all()))"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dfa.py tests/test_gpu_rploop.py tests/test_gpu_configs.py -m gpu -x -q -k "fragment or soak or pool or dfa or loop or cfg5 or walk" "$@" | tail -15; exit ${PIPESTATUS[0]}
