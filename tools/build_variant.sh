#!/bin/bash
# build a variant of the library for A/B runs (tools/ab.sh, CTC_AMD_LIB): recompiles ONE source of
# ctc_amd/csrc with extra -D flags and links it with the objects of the regular build
# usage: tools/build_variant.sh <name> <source.hip> [-DFLAG ...]   ->  ctc_amd/lib/variants/<name>.so
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python -m ctc_amd.build > /dev/null
mkdir -p "$ROOT/ctc_amd/lib/variants"
OBJ="$ROOT/ctc_amd/lib/variants/$NAME.o"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCTC_AMD_EXPERIMENTS "$@" -c "$ROOT/ctc_amd/csrc/$SRC" -o "$OBJ"
OTHERS=$(ls "$ROOT"/ctc_amd/lib/obj/*.o | grep -v "/${SRC%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ctc_amd/lib/variants/$NAME.so" "$OBJ" $OTHERS
echo "$ROOT/ctc_amd/lib/variants/$NAME.so"
