#!/bin/bash
# Build experiment variants of the tuning library: tools/build_variants.sh name "-DFLAG=..." [name flags ...]
# -> multimodal-vae-public_amd/libmvae_hip_tuning_<name>.so (used with MVAE_HIP_LIB=... tools/gemm_bench.py, bench.py,
# tools/ab_matrix.sh).  VARIANT_SRCS (default "linear conv norm poe") are rebuilt with the flags; the other objects are
# shared with the tuning build (VARIANT_SRCS=linear: a switch that only linear.hip reads -- a quarter of the build time).
set -e
cd "$(dirname "$0")/../multimodal-vae-public_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DMVAE_TUNING"
SRCS=${VARIANT_SRCS:-linear conv norm poe}
make -j8 >/dev/null
while [ $# -ge 2 ]; do
    name=$1; extra=$2; shift 2
    mkdir -p variants/$name
    objs=""
    for s in linear conv norm poe; do
        if echo " $SRCS " | grep -q " $s "; then
            ( $HIPCC $FLAGS $extra -c $s.hip -o variants/$name/$s.o 2>/dev/null ) &
            objs="$objs variants/$name/$s.o"
        elif [ -f tuning_$s.o ]; then objs="$objs tuning_$s.o"
        else objs="$objs $s.o"; fi
    done
    wait
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libmvae_hip_tuning_$name.so $objs \
        loss.o misc.o reparam.o gather.o preprocess.o comm.o gru.o -ldl
    echo built libmvae_hip_tuning_$name.so
done
