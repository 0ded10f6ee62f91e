#!/bin/bash
# ASan + UBSan pass over the host code under the C ABI: the script is test infrastructure (it also instruments the CPU checker) and lives in tests/.
exec bash "$(dirname "$0")/../tests/sanitize.sh" "$@"
