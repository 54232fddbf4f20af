#!/bin/bash
# ONE parameterised GPU-box session script (replaces the 47 one-shot tools/session_r0*.sh of rounds 4-5; what each of those
# ran is listed in tools/SESSIONS.md).  Run through gpurun from the repo root:
#
#   gpurun -- 'MVAE_GIT_HEAD=<head> bash tools/session.sh final r06'      the round's final collection (tables bench.py quotes)
#   gpurun -- 'bash tools/session.sh tests "adam or elbo"'                pytest -m gpu, optionally narrowed with -k
#   gpurun -- 'bash tools/session.sh ab 2 "celeba:lib:base,lib:variantA"' interleaved step A/B (tools/ab_matrix.sh)
#   gpurun -- 'bash tools/session.sh byshape r06x mnist celeba'           per-(call, shape) rocprofv3 tables of some workloads
#   gpurun -- 'bash tools/session.sh g2 lin|glin|conv [--sweep]'          version-2 GEMM core against the round-5 kernels
#
# Everything lands under gpurun_out/<mode>/ (scratch); copy what should be judged into profiles/.
set -u
cd "$GRAFT_REPO_ROOT"
mode=${1:-}; shift || true
TUNING=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning.so
case "$mode" in
final)
    export PFX=${1:-r06}
    out=gpurun_out/final; rm -rf $out; mkdir -p $out
    echo "head=${MVAE_GIT_HEAD:-unknown} pfx=$PFX" > $out/status.txt
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
    timeout 900 bash tools/collect_by_shape.sh > $out/by_shape.log 2>&1
    cp gpurun_out/by_shape/${PFX}_* $out/ 2>/dev/null
    cp gpurun_out/by_shape/${PFX}_by_shape.json profiles/${PFX}_by_shape.json 2>/dev/null
    TRAFFIC_TABLE=${PFX}_traffic.json timeout 900 bash tools/collect_traffic.sh "linear_fwd|M1024 N512 K512" "linear_wgrad_batched|4 layers" \
        "linear_dgrad|M1024 N512 K512" "convT2d_fwd|2048x64x14x14" "convT2d_dgrad|512x256x5x5" "convT2d_dgrad|2048x128x7x7" \
        "convT2d_fwd|4608x128x8x8" "convT2d_wgrad|256x128x4x4" > $out/traffic.log 2>&1
    cp gpurun_out/${PFX}_traffic.json $out/ 2>/dev/null; cp gpurun_out/${PFX}_traffic.json profiles/${PFX}_traffic.json 2>/dev/null
    t0=$(date +%s); timeout 900 python bench.py > $out/${PFX}_bench_default.json 2> $out/bench.err
    echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s" >> $out/status.txt
    timeout 900 bash tools/collect_profiles.sh > $out/collect_profiles.log 2>&1
    cp gpurun_out/profiles_new/${PFX}_* $out/ 2>/dev/null
    for w in mnist fashionmnist celeba celeba19; do
        timeout 200 python bench.py --workload $w --force-dp --no-extras > $out/dp_$w.json 2> $out/dp_$w.err; echo "dp $w rc=$?" >> $out/status.txt
        MVAE_GRID_REPORT=1 MVAE_HIP_LIB=$TUNING timeout 200 python tools/grid_report.py run $w 2> $out/grid_$w.err > /dev/null
        python tools/grid_report.py table $out/grid_$w.err > $out/grid_$w.txt; rm -f $out/grid_$w.err
    done
    cat $out/status.txt; tail -c 1500 $out/bench.err
    ;;
tests)
    out=gpurun_out/tests; mkdir -p $out
    if [ -n "${1:-}" ]; then python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -20 | tee $out/tests.txt
    else python -m pytest tests -m gpu -x -q 2>&1 | tail -20 | tee $out/tests.txt; fi
    ;;
ab)
    out=gpurun_out/ab; mkdir -p $out
    rounds=${1:-2}; shift || true
    timeout 2400 bash tools/ab_matrix.sh "$rounds" "$@" 2>&1 | tee $out/ab.txt
    ;;
byshape)
    export PFX=${1:-r06x}; shift || true
    timeout 1200 bash tools/collect_by_shape.sh "$@" > gpurun_out/by_shape.log 2>&1
    for f in gpurun_out/by_shape/${PFX}_*_by_shape.txt; do echo "== $f"; head -40 "$f" | cut -c1-200; done
    ;;
g2)
    out=gpurun_out/g2; mkdir -p $out
    python tools/g2_bench.py --cases "${1:-lin}" ${2:-} 2>&1 | tee $out/g2_bench_${1:-lin}.txt
    ;;
*)
    echo "usage: session.sh final|tests|ab|byshape|g2 ..." >&2; exit 2
    ;;
esac
