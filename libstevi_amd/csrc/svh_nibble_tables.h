// Constant tables of the matrix-core census sweep (svh_census_sweep_pm.hip) and of the kernels that prepare its operands
// (svh_features.hip): census bits as FP4 (e2m1) operands of v_mfma_scale_f32_32x32x64_f8f6f4.
//
//     bit 0 -> +1.0 (nibble 0x2),   bit 1 -> -1.0 (nibble 0xA),   for the source AND the target records
//     dot product over the B written bits of two records = (equal bits) - (differing bits) = B - 2 Hamming
//
// so the Hamming cost needs neither popcount: cost = (B - dot) / 2.  A record of 32 bits is 16 bytes: nibble e of dword q is bit
// 8 q + e (the k-order of the MFMA operand: lane half h of a fragment holds the k-subset [32 h, 32 h + 32)).
#pragma once

#include <cstdint>

namespace svh {

struct NibbleTables {
    uint32_t lut[256];      // byte -> its eight bits as eight nibbles
    uint32_t zero_vec[4];   // the 32 bits of an all-zero census word: what a target column outside the image holds (cross_correlations.h:235)
    uint32_t absent[4];     // a word that is not there (odd word count): 0.0 operands, contributes nothing
};

constexpr NibbleTables make_nibble_tables() {
    NibbleTables t{};
    for (unsigned b = 0; b < 256; b++) {
        uint32_t v = 0;
        for (unsigned e = 0; e < 8; e++) v |= (((b >> e) & 1u) ? 0xAu : 0x2u) << (4 * e);
        t.lut[b] = v;
    }
    for (int q = 0; q < 4; q++) {
        t.zero_vec[q] = 0x22222222u;
        t.absent[q] = 0u;
    }
    return t;
}

// one copy per translation unit that includes this header (1 KB each), constant-initialised in device memory
static __device__ const NibbleTables kNibbleTables = make_nibble_tables();

} // namespace svh
