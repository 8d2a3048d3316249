#!/usr/bin/env bash
# builds gpurun_libs/NAME.so = the library with extra compiler flags (ablation / parameter variants for tools/gpu_ab.sh)
# usage: tools/build_variant.sh NAME "-DFLAG ..."
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"; name="$1"; flags="${2:-}"
mkdir -p "$R/gpurun_libs" "$R/build/variants/$name/lib"
make -s -C "$R/cugraph_amd/csrc" -j8 OBJ="$R/build/variants/$name/obj" OUT="$R/build/variants/$name/lib" EXTRA="$flags" "$R/build/variants/$name/lib/libcugraph_c.so"
cp -f "$R/build/variants/$name/lib/libcugraph_c.so" "$R/gpurun_libs/$name.so"
echo "built gpurun_libs/$name.so ($flags)"
