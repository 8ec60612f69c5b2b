#!/bin/bash
# tools/build_variant.sh NAME "-DMACRO ..." — builds libear_amd/lib_NAME/libearhip.so with extra compiler flags
# (kernel variants for same-box A/B runs: EARHIP_LIB=libear_amd/lib_NAME/libearhip.so python bench.py ...)
set -e
cd /root/repo/libear_amd/csrc
make -s -j4 OUT=../lib_$1 EXTRA="$2" 2>&1 | grep -E "error|warning: " || true
ls -la ../lib_$1/libearhip.so
