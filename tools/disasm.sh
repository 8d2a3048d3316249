#!/usr/bin/env bash
# device ISA of one object of the library build: tools/disasm.sh spmv_tiled [kernel-name-regex] -> /tmp/<obj>.s (+ resource usage of the matching kernels)
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"; B=/opt/rocm/lib/llvm/bin
rm -rf "/tmp/disasm_$1"; mkdir -p "/tmp/disasm_$1"; cp "$R/build/obj/$1.o" "/tmp/disasm_$1/x.o"
(cd "/tmp/disasm_$1" && $B/llvm-objdump --offloading x.o > /dev/null)
co="/tmp/disasm_$1/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950"
$B/llvm-objdump -d "$co" > "/tmp/$1.s"
$B/llvm-readelf --notes "$co" | grep -E "\.name:|vgpr_count|sgpr_count|spill_count|group_segment" | paste - - - - - - | grep -E "${2:-.}" | sed 's/  */ /g'
