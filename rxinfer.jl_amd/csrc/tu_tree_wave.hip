// tu_tree_wave.hip — one dimension class of the executor's wavefront-per-item kernels: -DRXHIP_TU_DC=16 | 32 | 64 (register tiles: tree_tile_kernels.hpp; LDS-staged:
// tree_wave_kernels.hpp).
#include "tree_wave.hpp"

#ifndef RXHIP_TU_DC
#error "RXHIP_TU_DC: 16, 32 or 64"
#endif
#ifndef RXHIP_TILE_MAX
#define RXHIP_TILE_MAX 32   // dimension classes up to this one run the register-tile kernels (tree_tile_kernels.hpp), the ones above the LDS-staged kernels
#endif

#if RXHIP_TU_DC <= RXHIP_TILE_MAX
#include "tree_tile_kernels.hpp"
namespace rxhip {
namespace tree {
namespace wave {
namespace {
constexpr int NT = RXHIP_TU_DC / 16;
using namespace rxhip::tree::tile;

hipError_t prepare(int) { return hipSuccess; }
// (es = 1: the engines' element-fastest storage — scalar base + lane offset addressing; rxhip_rule_eval's one-node schedules are replica-fastest)
void ops(int phase, const TreeParams& p, int o0, int o1, int, unsigned blocks, hipStream_t stream) {
    if (p.es == 1) {
        if (phase == 0) hipLaunchKernelGGL((k_tile_ops<0, NT, true>), dim3(blocks), dim3(64), 0, stream, p, o0, o1);
        else if (phase == 1) hipLaunchKernelGGL((k_tile_ops<1, NT, true>), dim3(blocks), dim3(64), 0, stream, p, o0, o1);
        else if (phase == 2) hipLaunchKernelGGL((k_tile_ops<2, NT, true>), dim3(blocks), dim3(64), 0, stream, p, o0, o1);   // (2: the second phase's light opcodes)
        else hipLaunchKernelGGL((k_tile_ops<3, NT, true>), dim3(blocks), dim3(64), 0, stream, p, o0, o1);                    // (3: OP_FE_NOISE2M alone)
    } else
        hipLaunchKernelGGL((k_tile_ops<0, NT, false>), dim3(blocks), dim3(64), 0, stream, p, o0, o1);   // (the sweep's rules only)
}
void walk(int phase, const TreeParams& p, int o0, int o1, int, unsigned blocks, hipStream_t stream) {
    if (phase == 0) hipLaunchKernelGGL((k_tile_walk<0, NT, true>), dim3(blocks), dim3(64), 0, stream, p, o0, o1);
    else hipLaunchKernelGGL((k_tile_walk<1, NT, true>), dim3(blocks), dim3(64), 0, stream, p, o0, o1);
}
const WaveVtbl VT = {prepare, ops, walk};
}  // namespace
#else
#include "tree_wave_kernels.hpp"

namespace rxhip {
namespace tree {
namespace wave {
namespace {
constexpr int DC = RXHIP_TU_DC;

hipError_t prepare(int dmax) {
    const int bytes = (int)lds_bytes(dmax);
    if (bytes <= 64 * 1024) return hipSuccess;
    for (const void* f : {(const void*)k_wave_ops<0, DC>, (const void*)k_wave_ops<1, DC>, (const void*)k_wave_walk<0, DC>, (const void*)k_wave_walk<1, DC>})
        if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) return e;
    return hipSuccess;
}
void ops(int phase, const TreeParams& p, int o0, int o1, int dmax, unsigned blocks, hipStream_t stream) {
    if (phase >= 2) phase = 1;   // (one second-phase instance here)
    if (phase == 0) hipLaunchKernelGGL((k_wave_ops<0, DC>), dim3(blocks), dim3(WL), lds_bytes(dmax), stream, p, o0, o1, dmax);
    else hipLaunchKernelGGL((k_wave_ops<1, DC>), dim3(blocks), dim3(WL), lds_bytes(dmax), stream, p, o0, o1, dmax);
}
void walk(int phase, const TreeParams& p, int o0, int o1, int dmax, unsigned blocks, hipStream_t stream) {
    if (phase == 0) hipLaunchKernelGGL((k_wave_walk<0, DC>), dim3(blocks), dim3(WL), lds_bytes(dmax), stream, p, o0, o1, dmax);
    else hipLaunchKernelGGL((k_wave_walk<1, DC>), dim3(blocks), dim3(WL), lds_bytes(dmax), stream, p, o0, o1, dmax);
}
const WaveVtbl VT = {prepare, ops, walk};
}  // namespace
#endif

#if RXHIP_TU_DC == 16
const WaveVtbl* wave_vt16() { return &VT; }
#elif RXHIP_TU_DC == 32
const WaveVtbl* wave_vt32() { return &VT; }
#else
const WaveVtbl* wave_vt64() { return &VT; }
#endif

}  // namespace wave
}  // namespace tree
}  // namespace rxhip
