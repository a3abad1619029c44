#!/bin/bash
# A second build of the library with extra compiler flags on one source file, for timing experiments:
#   tools/build_variant.sh <name> <file.hip> <flags...>   -> graphsage_amd/variants/libgsage_<name>.so
# (run with GSAGE_LIB=graphsage_amd/variants/libgsage_<name>.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
cd $ROOT/graphsage_amd/csrc
make -s
mkdir -p ../variants build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wall -Wno-unused-function "$@" -c $src -o build/var/$name.o
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o" | grep -v "_safewait.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libgsage_$name.so $objs build/var/$name.o -Wl,-rpath,/opt/rocm/lib
echo "built variants/libgsage_$name.so"
