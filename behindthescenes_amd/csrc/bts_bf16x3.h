// bts_bf16x3.h -- fp32 products on the bf16 matrix pipe, exactly.
//
// An fp32 number is the sum of three bf16 numbers: h = its top 16 bits, m = the top 16 bits of x - h, l = x - h - m (8 + 8 + 8 significand
// bits; all three subtractions are exact, and bf16 carries fp32's exponent: nothing to scale), so
//     x w = xh wh + (xh wm + xm wh) + (xm wm + xh wl + xl wh) + [xm wl + xl wm + xl wl: below 2^-24 |x w|, dropped]
// is six v_mfma_f32_32x32x16_bf16 (8 passes, k = 16) where the fp32-input pipe needs eight v_mfma_f32_32x32x2_f32 (16 passes, k = 2
// each): 0.375 of the matrix time, every product exact in the fp32 accumulator; measured against fp64 the sums are CLOSER than the
// fp32-input MFMA's (it adds 16 products at a time).  The price is the split: 5.5 VALU instructions per operand element.
// Used by the decoder tail's convolutions (bts_conv.hip).  Tried for the sparse projection backward too and not kept: that pass is
// traffic-bound, a third of its matrix time moves it by 4 % (profiles/r05u).
#pragma once
#include "bts_common.h"

namespace bts {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }
// (a, b) -> the three bf16 pairs (a in the low half): h, m, l
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
  const float ra = a - u2f(f2u(a) & 0xFFFF0000u), rb = b - u2f(f2u(b) & 0xFFFF0000u);
  const float la = ra - u2f(f2u(ra) & 0xFFFF0000u), lb = rb - u2f(f2u(rb) & 0xFFFF0000u);
  ph = __builtin_amdgcn_perm(f2u(b), f2u(a), 0x07060302u);
  pm = __builtin_amdgcn_perm(f2u(rb), f2u(ra), 0x07060302u);
  pl = __builtin_amdgcn_perm(f2u(lb), f2u(la), 0x07060302u);
}
// a lane's 8 consecutive channels -> its A (or B) fragment of a k-step, three terms
__device__ __forceinline__ void split3_frag(const float4& v0, const float4& v1, bf8& fh, bf8& fm, bf8& fl) {
  unsigned h[4], m[4], l[4];
  split3_pair(v0.x, v0.y, h[0], m[0], l[0]), split3_pair(v0.z, v0.w, h[1], m[1], l[1]);
  split3_pair(v1.x, v1.y, h[2], m[2], l[2]), split3_pair(v1.z, v1.w, h[3], m[3], l[3]);
  fh = __builtin_bit_cast(bf8, (u32x4c){h[0], h[1], h[2], h[3]}), fm = __builtin_bit_cast(bf8, (u32x4c){m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(bf8, (u32x4c){l[0], l[1], l[2], l[3]});
}

struct WSplit {
  float v, r1, r2;   // x, x - h, x - h - m: the top 16 bits of the three are the bf16 terms
};
__device__ __forceinline__ WSplit wsplit(float x) {
  WSplit o;
  o.v = x;
  o.r1 = x - u2f(f2u(x) & 0xFFFF0000u);
  o.r2 = o.r1 - u2f(f2u(o.r1) & 0xFFFF0000u);
  return o;
}
template <int N>
__device__ __forceinline__ void wfrags(const WSplit (&sv)[N], int first, bf8& fh, bf8& fm, bf8& fl) {   // pairs (first + 2 j, first + 2 j + 1), j = 0..3
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const WSplit &a = sv[first + 2 * j], &b = sv[first + 2 * j + 1];
    h[j] = __builtin_amdgcn_perm(f2u(b.v), f2u(a.v), 0x07060302u);
    m[j] = __builtin_amdgcn_perm(f2u(b.r1), f2u(a.r1), 0x07060302u);
    l[j] = __builtin_amdgcn_perm(f2u(b.r2), f2u(a.r2), 0x07060302u);
  }
  fh = __builtin_bit_cast(bf8, (u32x4c){h[0], h[1], h[2], h[3]}), fm = __builtin_bit_cast(bf8, (u32x4c){m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(bf8, (u32x4c){l[0], l[1], l[2], l[3]});
}

// the six products of one k-step on one accumulator, smallest terms first: acc += A . B with A = (ah, am, al), B = (bh, bm, bl)
__device__ __forceinline__ void mfma6_ab(f32x16& acc, const bf8& ah, const bf8& am, const bf8& al, const bf8& bh, const bf8& bm, const bf8& bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

}  // namespace bts
