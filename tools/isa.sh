#!/bin/bash
# tools/isa.sh <file.hip> <kernel-name-regex> [EXTRA flags]: gfx950 ISA of one kernel (comments stripped) on stdout
src=$1; pat=$2; shift 2
cd "$(dirname "$0")/../uni-slam_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fPIC -x hip -S --cuda-device-only "$@" -o /tmp/isa_$$.s "$src" 2> >(grep -v hip-link >&2) || exit 1
awk -v pat="^$pat" '$0 ~ pat && /:/ {on=1} on {print} on && /s_endpgm/ {exit}' /tmp/isa_$$.s | grep -v '^\s*;'
grep -A12 "\.name: *$pat" /tmp/isa_$$.s | grep "name:\|sgpr_count\|vgpr_count\|spill" >&2
grep -B12 "\.name: *$pat" /tmp/isa_$$.s | grep "group_segment_fixed_size" >&2
rm -f /tmp/isa_$$.s
