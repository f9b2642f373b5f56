#!/bin/bash
# tools/build_variant.sh <suffix> <file.hip> [-D...]: apertis_llm_amd/libapertis_hip_<suffix>.so = the in-tree library with ONE
# source rebuilt under extra -D switches (A/B runs: APERTIS_HIP_LIB=...; never shipped - the name is git-ignored)
set -e
cd "$(dirname "$0")/.."
SFX=$1; SRC=$2; shift 2
python -m apertis_llm_amd.build > /dev/null
O=apertis_llm_amd/csrc/_obj
mkdir -p $O/_var
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result "$@" -c apertis_llm_amd/csrc/$SRC -o $O/_var/${SRC%.hip}_$SFX.o
mkdir -p $O/_var
OBJS=$(ls $O/*.o | grep -v "/${SRC%.hip}\.o$")
hipcc -shared -fPIC --offload-arch=gfx950 -o apertis_llm_amd/libapertis_hip_$SFX.so $OBJS $O/_var/${SRC%.hip}_$SFX.o
rm -f $O/_var/${SRC%.hip}_$SFX.o
echo apertis_llm_amd/libapertis_hip_$SFX.so
