#!/bin/bash
# build_variant.sh NAME "-DFLAG=1 ..." [PATTERN] : builds tools/lib_NAME.so, an experiment library in which the sources
# matching PATTERN (default: every .hip file) are compiled with the extra flags; load it with TMDIFF_HIP_LIB.
set -e
cd "$(dirname "$0")/.."
pat=${3:-.hip}
mkdir -p build/var_$1
for f in tmdiff_amd/csrc/abi.cpp tmdiff_amd/csrc/*.hip; do
  o=build/var_$1/$(basename $f).o
  if [[ $f == *$pat* || ! -f build/$(basename $f).o ]]; then
    /opt/rocm/bin/hipcc $2 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Itmdiff_amd/csrc -x hip -c $f -o $o
  else
    cp build/$(basename $f).o $o
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/var_$1/*.o -o tools/lib_$1.so
