#!/bin/bash
# tools/mkvariant.sh <name> <tu.hip> <flags...>: build/ab/<name>.so = the product library with ONE translation unit rebuilt with extra
# flags (A/B timing variants for tools/ab.sh; the other objects are taken from chirpgp_amd/csrc as built by make)
set -e
NAME=$1; TU=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/chirpgp_amd/csrc"
mkdir -p "$ROOT/build/ab"
EXTRA=$(make -s -n -W $TU ${TU%.hip}.o 2>/dev/null | grep -o -- '-mllvm [^ ]*' | tr '\n' ' ')
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -Wno-int-to-pointer-cast -I../../include \
    $EXTRA "$@" -c $TU -o "$ROOT/build/ab/$NAME.o" 2> >(grep -v hip-link >&2)
OBJS=$(ls *.o | grep -v "^${TU%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/build/ab/$NAME.so" $OBJS "$ROOT/build/ab/$NAME.o" 2> >(grep -v hip-link >&2)
echo "built build/ab/$NAME.so ($EXTRA $@)"
