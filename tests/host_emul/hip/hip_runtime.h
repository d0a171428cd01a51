// tests/host_emul/hip/hip_runtime.h — NOT the HIP runtime: the handful of names the executor's kernel headers use, defined for a host compiler, so that
// the rule bodies of csrc/tree_kernels.hpp (registers, one lane) and csrc/tree_wave_kernels.hpp (LDS-staged, one wavefront) can be run against each other
// by g++ without a GPU (tests/test_tree_wave_host.py).  Only tests/ puts this directory on an include path.
#pragma once
#include <cmath>
#include <cstddef>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct emul_dim3 { unsigned x = 0, y = 0, z = 0; };
static emul_dim3 blockIdx, threadIdx, blockDim, gridDim;
static inline void __syncthreads() {}
static inline int atomicOr(int* p, int v) { const int o = *p; *p |= v; return o; }
