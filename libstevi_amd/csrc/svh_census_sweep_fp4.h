// Shared by the two FP4 matrix-core census sweeps (svh_census_sweep_pm.hip: any geometry; svh_census_sweep_rl.hip: RightToLeft with
// the Pout boundary on the target image's right edge, compile-time disparity count): operand encoding, accumulator conventions,
// the walk of persistent blocks over items.
#pragma once

#include <algorithm>

#include "svh_census_keys.h"

namespace svh {
namespace fp4sweep {

using v8i = int __attribute__((ext_vector_type(8)));
using v16f = float __attribute__((ext_vector_type(16)));

constexpr float BIG_CELL = 4194304.0f;        // an invalid cell starts this far below: never the maximum of a tile that has a valid one
constexpr float BIG_G = 268435456.0f;           // "no candidate yet" is -BIG_G
constexpr float G_VALID_ABOVE = -1048576.0f;  // valid packed maxima stay above

__device__ __forceinline__ constexpr int row_of_reg(int reg) { return (reg & 3) + 8 * (reg >> 2); } // + 4 * (lane >> 5)

// max of three (v_max3_f32).  Not inline asm: the tree is the first reader of the MFMA result and the compiler only pads the
// MFMA -> VALU read hazard for instructions it can see.
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// The 32 bits of a census word as the 32 FP4 operands of one lane-half k-subset, 16 bytes: bit 0 -> +1.0 (nibble 0x2), bit 1 -> -1.0
// (0xA), i.e. nibble = 0x2 | bit << 3.  Dword q of the record holds bits q, q + 4, ..., q + 28 in its nibbles 0 .. 7 -- not the bits in
// order, which a dot product does not care about as long as both operands use the same order; both come from this function.  With
// that order a dword is ((w << (3 - q)) & 0x88888888) | 0x22222222: a shift and a v_and_or_b32, 7 VALU instructions per word (the
// first form, bit pairs looked up by v_perm_b32 in a four-entry byte table, took 11; a 256-entry byte table in LDS 8 + four
// conflict-ridden ds_read_b32).
// (Inline asm: the compiler splits (x & m) | c with two literal constants into v_and + v_or, 11 instructions again; with the mask in a
// scalar register and the ones in a vector register it is one VOP3.)
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t mask_sgpr, uint32_t ones_vgpr) {
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(mask_sgpr), "v"(ones_vgpr));
    return d;
}
__device__ __forceinline__ uint4 word_record(uint32_t w) {
    constexpr uint32_t SIGNS = 0x88888888u, ONES = 0x22222222u;
    return make_uint4(and_or(w << 3, SIGNS, ONES), and_or(w << 2, SIGNS, ONES), and_or(w << 1, SIGNS, ONES), and_or(w, SIGNS, ONES));
}

// 16 bytes per lane from global memory straight into LDS: lane l lands at lds_wave_base + 16 l (the base is wave uniform)
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds_wave_base,
                                     16, 0, 0);
}

struct SweepPlan {   // host-side decomposition of a launch (make_plan)
    int bpr;         // items per image row
    int items;       // items per XCD list: ceil(H / 8) * bpr (rows i = 8 q + xcd)
    int step;        // blocks per XCD list = gridDim.x / 8: a block takes items slot, slot + step, ...
    int step_q, step_j; // step = step_q * bpr + step_j: the same walk without a division per item
    int nbuf;        // window buffers in LDS: 2 (the expansion of item n + 1 overlaps other waves' tiles of item n) or 1
};

// the walk of persistent blocks: a multiple of 8 blocks (every XCD list gets the same number of walkers), `per_cu` per CU at most
inline SweepPlan make_walk(int H, int Ws, int pxb, int n_cu, int per_cu, int nbuf, int *grid) {
    SweepPlan p;
    p.bpr = ceil_div(Ws, pxb);
    p.items = ceil_div(H, 8) * p.bpr;
    p.nbuf = nbuf;
    int g = n_cu * std::max(per_cu, 1);
    g = std::min(g, p.items * 8);
    g = std::max(8, (g / 8) * 8);
    p.step = g / 8;
    p.step_q = p.step / p.bpr;
    p.step_j = p.step % p.bpr;
    *grid = g;
    return p;
}

} // namespace fp4sweep
} // namespace svh
