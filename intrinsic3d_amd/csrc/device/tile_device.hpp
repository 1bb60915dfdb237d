// Register / plan-word helpers shared by the single-system operator pass (tile_pass.hip) and the multi-system one (tile_pass_mr.hip).
#pragma once
#include "common.hpp"

namespace i3d {

struct RowBlock { float4 p[7]; float j28; int tag; };  // one stored Eg row in registers: planes 0..6 + column 28 + keyframe tag
typedef unsigned v4u_b __attribute__((ext_vector_type(4)));
typedef unsigned v2u_b __attribute__((ext_vector_type(2)));
// local slot j of an entry's packed plan words (12 bits each, LSB first; j is a compile-time constant after unrolling: one v_bfe_u32, or v_alignbit + v_and
// for the slots that straddle a word)
template <int NW> static __device__ inline int unpack12(const unsigned (&w)[NW], int j) {
    const int bit = 12 * j, k = bit >> 5, sh = bit & 31;
    if (sh <= 20) return (int)((w[k] >> sh) & 0xFFFu);
    return (int)(((w[k] >> sh) | (w[k + 1 < NW ? k + 1 : k] << (32 - sh))) & 0xFFFu);
}
template <int Z> struct AllZ { unsigned w[LNBR_WORDS]; constexpr AllZ() : w{} { for (int j = 0; j < 18; ++j) { const int bit = 12 * j, k = bit >> 5, sh = bit & 31; w[k] |= (unsigned)Z << sh; if (sh > 20) w[k + 1] |= (unsigned)Z >> (32 - sh); } } };


}  // namespace i3d
