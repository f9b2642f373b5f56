#!/bin/bash
# usage: tools/isa_loop.sh <mangled-kernel-prefix> [lines]  - compile grouped_gemm.hip to ISA and print the MFMA region's issue order
set -e
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -S --cuda-device-only -o /tmp/gg.s /root/repo/apertis_llm_amd/csrc/grouped_gemm.hip 2>&1 | grep " error" || true
grep -n "vgpr_spill_count\|\.name:  " /tmp/gg.s | grep -A1 "$1" | head -4
L=$(grep -n "^$1.*:" gg.s | head -1 | cut -d: -f1)
sed -n "${L},\$p" gg.s | awk '/s_endpgm/{print; exit} {print}' > ntp.s
S=$(grep -n "v_mfma" ntp.s | head -1 | cut -d: -f1)
sed -n "$((S-12)),$((S+${2:-160}))p" ntp.s | grep -v "^\s*;" | awk '{print $1, $2, $3}' | awk '{ if ($1 ~ /v_mfma/) {m++; if (last!="m") {printf "\n"}; printf "M "; last="m"} else { if (last=="m") printf "\n"; print "   " $0; last="o"} }'
