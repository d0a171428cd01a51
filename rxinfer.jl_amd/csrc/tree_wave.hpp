// tree_wave.hpp — the seam between the executor's host side (tree_engine.hip) and the LDS-staged kernels for dimensions above 8 (tree_wave_kernels.hpp):
// one translation unit per dimension class (tu_tree_wave.hip, -DRXHIP_TU_DC=16 | 32 | 64: dmax ≤ 16, ≤ 32, ≤ 64), built side by side.
#pragma once
#include <hip/hip_runtime.h>

#include "tree_kernels.hpp"

namespace rxhip {
namespace tree {
namespace wave {

struct WaveVtbl {
    hipError_t (*prepare)(int dmax);   // dynamic LDS above the default 64 KB limit (per device: the attribute lives with the loaded code object)
    // phase 0: the sweep, 1: Bethe terms / q(W) updates.  ops: a wavefront per (op, replica) of [o0, o1) (one level); walk: a wavefront per replica over [o0, o1)
    void (*ops)(int phase, const TreeParams& p, int o0, int o1, int dmax, unsigned blocks, hipStream_t stream);
    void (*walk)(int phase, const TreeParams& p, int o0, int o1, int dmax, unsigned blocks, hipStream_t stream);
};
const WaveVtbl* wave_vt16();
const WaveVtbl* wave_vt32();
const WaveVtbl* wave_vt64();
inline const WaveVtbl* wave_vt(int dmax) { return dmax <= 16 ? wave_vt16() : dmax <= 32 ? wave_vt32() : wave_vt64(); }
constexpr int DMAX_WAVE = 64;

}  // namespace wave
}  // namespace tree
}  // namespace rxhip
