#!/bin/bash
# Build experiment variants of the tuning library: tools/build_variants.sh name "-DFLAG=..." [name flags ...]
# -> multimodal-vae-public_amd/libmvae_hip_tuning_<name>.so (used with MVAE_HIP_LIB=... tools/gemm_bench.py, bench.py,
# tools/ab_matrix.sh).  linear.hip, conv.hip, norm.hip and poe.hip are rebuilt with the flags; the other objects are shared.
set -e
cd "$(dirname "$0")/../multimodal-vae-public_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DMVAE_TUNING"
make -j8 >/dev/null
while [ $# -ge 2 ]; do
    name=$1; extra=$2; shift 2
    mkdir -p variants/$name
    ( $HIPCC $FLAGS $extra -c linear.hip -o variants/$name/linear.o 2>/dev/null ) &
    ( $HIPCC $FLAGS $extra -c conv.hip -o variants/$name/conv.o 2>/dev/null ) &
    ( $HIPCC $FLAGS $extra -c norm.hip -o variants/$name/norm.o 2>/dev/null ) &
    ( $HIPCC $FLAGS $extra -c poe.hip -o variants/$name/poe.o 2>/dev/null ) &
    wait
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libmvae_hip_tuning_$name.so variants/$name/linear.o variants/$name/conv.o \
        variants/$name/norm.o variants/$name/poe.o loss.o misc.o reparam.o gather.o preprocess.o comm.o gru.o -ldl
    echo built libmvae_hip_tuning_$name.so
done
