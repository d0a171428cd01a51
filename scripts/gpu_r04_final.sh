#!/bin/bash
# round end: the -m gpu suite + one bench line, the profiles of the same build, and the per-chain element kernels on their own
set -u
TAG=${1:-r04h}
./scripts/gpu_r04_full.sh
./scripts/profile_r04.sh "$TAG"
F='RCCL\|HIP ver\|ROCm\|Hostname\|Librccl'
timeout 300 python scripts/time_per_chain_elements.py 2>&1 | grep -v "$F" | tee gpurun_out/prof_$TAG/per_chain_elements.txt
cat gpurun_out/prof_$TAG/c1_breakdown.txt
