#!/bin/bash
# CPU-side sanitizer run ("host code under -fsanitize=address,undefined"; GPU AddressSanitizer is not
# available on this pool).  Two targets, both without a GPU:
#   1. the oracle's C restatement (gcc -fsanitize=address,undefined) driven through every CPU parity test of tests/
#   2. the HOST side of libpdx_hip.so (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: argument checks, the Arrow IPC
#      parser / writer incl. the garbage, truncated and crafted streams of tests/test_ipc.py, the Parquet footer parser incl. the
#      refused files and 300 single-byte footer corruptions of tests/test_parquet.py, the ABI symbol table) loaded in place of
#      the product build through PDX_LIB_PATH
# Usage: tools/sanitize_cpu.sh   (writes tools/_san/, prints the pytest summaries; exit 0 = no sanitizer report)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/_san"
rm -rf "$OUT" && mkdir -p "$OUT"
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"

echo "== 1. oracle (gcc ASan + UBSan)"
gcc -O1 -g -fPIC -std=c11 -Wall -Wextra -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o "$OUT/libpdx_oracle.so" "$ROOT/oracle/pdx_oracle.c" -lm
GCC_ASAN="$(gcc -print-file-name=libasan.so)"
( cd "$ROOT" && PDX_ORACLE_SO="$OUT/libpdx_oracle.so" LD_PRELOAD="$GCC_ASAN" \
    python -m pytest tests/test_oracle_golden.py tests/test_oracle_golden_r2.py tests/test_oracle_golden_r3.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3 )

echo "== 2. libpdx_hip.so host side (clang ASan + UBSan, device code unsanitised)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CSRC="$ROOT/pandasarrow_amd/csrc"
OBJS=()
for f in "$CSRC"/*.hip; do
  o="$OUT/$(basename "${f%.hip}").o"
  "$HIPCC" -O1 -g --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
      -Wno-unused-value -Wno-unused-function -Wno-cuda-compat -I"$ROOT/include" -I"$CSRC" -c "$f" -o "$o" &
  OBJS+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o "$OUT/libpdx_hip.so" "${OBJS[@]}" -ldl
CLANG_ASAN="$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)"
( cd "$ROOT" && PDX_LIB_PATH="$OUT/libpdx_hip.so" LD_PRELOAD="$CLANG_ASAN" \
    python -m pytest tests/test_ipc.py tests/test_parquet.py tests/test_abi_symbols.py -x -q -m "not gpu" -k "not built_from_these_sources" -p no:cacheprovider 2>&1 | tail -3 )
echo "sanitizer run finished without reports"
