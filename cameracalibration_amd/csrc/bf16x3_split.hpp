// bf16x3_split.hpp -- the exact three-way bf16 split of FP32 operands for v_mfma_f32_16x16x32_bf16
// (shared by the dense residual kernels and the RANSAC bracket pass).
//     v = v_h + v_m + v_l   (8 + 8 + 8 significand bits, exact for any float)
// The eight significant cross products of a coordinate fill one lane group's 8 K-slots (the dropped
// v_l*w_l term is below 2^-32 relative), products of bf16 pairs are exact in FP32, and the pipe
// accumulates them in FP32:
//   lane group g < 3 (coordinate g):  A = [wh wh wm wh wm wl wm wl]
//                                     B = [ph pm ph pl pm ph pl pm]
//   lane group 3 (the constant, p=1): A = [w3h 0 w3m 0 0 w3l 0 0], B = same split of 1.0
#pragma once

#include "kernels_common.hpp"

namespace ccal {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void split_bf16x3(float v, unsigned& h, unsigned& m, unsigned& l)
{
    h = __float_as_uint(v) & 0xffff0000u;             // truncation keeps every piece exact
    const float r1 = v - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2);      // what is left has at most 8 significant bits: already a bf16 value
}

// two bf16 given as FP32 bit patterns with zero low halves: low 16 bits = a, high = b
__device__ inline unsigned pack_bf16(unsigned a, unsigned b) { return (a >> 16) | b; }

__device__ inline bf16x8 split_point_operand(float v)
{
    unsigned h, m, l;
    split_bf16x3(v, h, m, l);
    const u32x4 r = {pack_bf16(h, m), pack_bf16(h, l), pack_bf16(m, h), pack_bf16(l, m)};
    return __builtin_bit_cast(bf16x8, r);
}

__device__ inline bf16x8 split_candidate_operand(float w, bool constant_group)
{
    unsigned h, m, l;
    split_bf16x3(w, h, m, l);
    u32x4 r;
    if (constant_group) r = u32x4{pack_bf16(h, 0u), pack_bf16(m, 0u), pack_bf16(0u, l), 0u};
    else r = u32x4{pack_bf16(h, h), pack_bf16(m, h), pack_bf16(m, l), pack_bf16(m, l)};
    return __builtin_bit_cast(bf16x8, r);
}

} // namespace ccal
