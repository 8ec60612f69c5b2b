#!/bin/bash
# CPU-side AddressSanitizer + UBSan pass (GPU sanitizers are not available on the
# pool): the oracle behind its test-suite, and the product's host-only sources
# (VariableBlockSizeAdapter FIFO, decorrelator design) behind a small driver.
# Usage: bash tools/sanitize.sh      (from the repo root; needs no GPU)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/earhip_sanitize
mkdir -p "$OUT"
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared"
g++ -std=c++14 $SAN -ffp-contract=off -o "$OUT/liboracle.so" "$ROOT/oracle/oracle_capi.cpp"
cat > "$OUT/stub.cpp" <<'EOS'
#include <string>
namespace earhip { static thread_local std::string g; void set_last_error(const std::string &s) { g = s; } }
extern "C" const char *earhip_last_error() { return earhip::g.c_str(); }
// (device-reachable host memory lives in api_core.hip; the FIFO under test here uses ordinary memory)
struct earhip_ctx;
extern "C" int earhip_host_alloc(earhip_ctx *, size_t, void **) { return 3; }
extern "C" int earhip_host_release(earhip_ctx *, void *) { return 3; }
EOS
g++ -std=c++17 $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" \
    -o "$OUT/libhost.so" "$ROOT/libear_amd/csrc/api_vbs.cpp" "$ROOT/libear_amd/csrc/api_decorrelate.cpp" "$OUT/stub.cpp"
# python does not link libstdc++, so ASan's __cxa_throw interceptor needs it preloaded
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so)"
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1
export EARHIP_SANITIZE_DIR="$OUT"
cd "$ROOT"
python3 tools/sanitize_oracle_tests.py
python3 tools/sanitize_host.py
echo "sanitize: clean"
