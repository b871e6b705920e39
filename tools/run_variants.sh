#!/bin/bash
# runs conv_fit.py bf16 with the product library and each tools/lib_*.so experiment library
cd "$(dirname "$0")/.."
echo "== product"; timeout -k 10 120 python tools/conv_fit.py bf16 2>&1 | grep "co=64"
for l in tools/lib_*.so; do echo "== $l"; TMDIFF_HIP_LIB=$PWD/$l timeout -k 10 120 python tools/conv_fit.py bf16 2>&1 | grep "co=64"; done
