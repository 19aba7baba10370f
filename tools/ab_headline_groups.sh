#!/bin/bash
# Headline scene (484 spheres, k_trace_mfma32) against the multi-level filter of k_trace_mfma_tiled (RT3_FORCE_TILED=1) at several leaf / row sizes.
# Libraries: build_ab/librt3hip_g<G>s<SUP>.so = -DRT3_GROUP_SPH=G -DRT3_SUPER=SUP; the default library is G = 8, SUP = 8.
out=${1:-gpurun_out/ab_headline_groups.log}
mkdir -p "$(dirname "$out")"
: > "$out"
for lib in raytracer-3_amd/librt3hip.so build_ab/librt3hip_g2s8.so build_ab/librt3hip_g4s4.so build_ab/librt3hip_g2s4.so build_ab/librt3hip_g4s8.so; do
    echo "== $lib" >> "$out"
    RT3_LIB_PATH=$PWD/$lib timeout -k 10 240 python tools/ab_env.py RT3_FORCE_TILED 64 3 >> "$out" 2>&1 || echo "FAILED $lib" >> "$out"
done
cat "$out"
