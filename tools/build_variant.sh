#!/bin/bash
# Builds a variant of libddsp_hip.so in which ONE source file is recompiled with extra flags (ablation / A-B builds):
#   tools/build_variant.sh <source.hip> <output suffix> <extra flags...>   ->  ddsp-pytorch_amd/libddsp_hip_<suffix>.so
# e.g. tools/build_variant.sh ddsp_noise_wave.hip abl1 -DDDSP_NOISE_ABL=1 ; run with DDSP_HIP_LIB=$PWD/ddsp-pytorch_amd/libddsp_hip_abl1.so
set -e
SRC=$1; SUF=$2; shift 2
cd "$(dirname "$0")/../ddsp-pytorch_amd/csrc"
make -s -j8
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -fno-slp-vectorize -Wall -Wno-unused-function -I../../include"
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$SRC" -o "/tmp/variant_$SUF.o"
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../libddsp_hip_$SUF.so" $OBJS "/tmp/variant_$SUF.o"
echo "built ddsp-pytorch_amd/libddsp_hip_$SUF.so"
