// Backward of the fused renderer, lane = SAMPLE form: three passes for the plain MLP (no ResnetBlockFC layers) with K <= 64.
//
// The round-1 backward (bts_bwd.hip) walks 64 rays per wave back to front, lane = ray: 392 registers and 138 KB of LDS tiles pin it at
// one wave per SIMD with every latency exposed, and 1 GB of g_h rows travels between its two passes (2.3 ms at 65 536 x 64).  The
// gradient at lin_in's output factorises,  g_h[p][ch] = [h[p][ch] > 0] * w_out[ch] * g_s[p],  so what the passes have to hand each
// other per sample is ONE float (g_s, the gradient at the pre-softplus density) and ONE BIT per channel (the relu gate):
//   pass A  rows_kernel      one ray per wave iteration, lane = sample -- the FORWARD's pipeline (f16-split lin_in on the matrix pipe,
//                            gather blended between the encoding regions, 2 waves / SIMD), so h and with it every relu gate is
//                            bit-identical to what the forward evaluated.  The compositing gradient is a suffix scan across the
//                            lanes.  Writes g_s and the gate masks (12 bytes per sample instead of a 256-byte row), and reduces
//                            dw_out = sum relu(h) g_s (the one term that needs h itself) and db_out = sum g_s on the spot.
//   pass B  scatter_kernel   one wave per (8x8 patch, 32-channel half): tap updates w_tap * g_s * [gate] * w_out merged per texel in a
//                            sliding LDS window, two points per read-modify-write round; d_empty from a dedicated window row.
//   pass C  dwpe_kernel      dW_pe^T[ch][kin] = sum_p [gate] w_out[ch] * (g_s pe)[p][kin]: one ray per wave iteration, the encoding
//                            recomputed lane = sample, scaled by g_s and transposed through a per-wave LDS tile; fp32 MFMA.
// What torch.autograd would do for nerf.py:283-299 + models_bts.py:266-338 + resnetfc.py:132-184 of the reference.
#define BTS_NO_LAUNCH_GLUE
#include "bts_render_kernel.h"
#include "bts_bwd.h"
#include <cstdlib>
#include <type_traits>

namespace bts {

// exclusive suffix sum over the wave: out[l] = sum_{m > l} x[m]
__device__ __forceinline__ float wave_suffix_excl(float x, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float y = __shfl_down(x, off, 64);
    x += (lane + off < 64) ? y : 0.0f;
  }
  const float s = __shfl_down(x, 1, 64);
  return lane == 63 ? 0.0f : s;
}

// sum over the 32 lanes of each wave half of 16 registers at once (butterfly with register halving: 2 selects, one shuffle and one
// add per surviving register and step).  Afterwards both lanes of the pair (col, col ^ 1) hold the total of register col >> 1
__device__ __forceinline__ float half_transpose_reduce16(float (&w)[16], int col) {
  int d = 16;
#pragma unroll
  for (int n = 16; n > 1; n >>= 1) {
    const bool up = (col & d) != 0;
#pragma unroll
    for (int j = 0; j < n / 2; ++j) {
      // opaque values: otherwise the two selects become w[up ? .. : ..], a dynamically indexed array (16-deep select chains)
      float lo = w[j], hi = w[n / 2 + j];
      asm("" : "+v"(lo), "+v"(hi));
      const float keep = up ? hi : lo;
      const float give = up ? lo : hi;
      w[j] = keep + __shfl_xor(give, d, 64);
    }
    d >>= 1;
  }
  return w[0] + __shfl_xor(w[0], 1, 64);
}

// Lanes L0 / L1 of (v0, v1) take the wave-uniform pairs (x00, x01) / (x10, x11)  (v_writelane_b32 with constant lane selects).
// gfx950 does NOT interlock a VALU read of an SGPR that a VALU instruction wrote within the previous two wait states (LLVM's
// VALUWriteSGPRVALURead hazard); hipcc pads its own instructions but does not look into inline assembly, and the values written here
// are fresh v_cmp results -- without the s_nop 1.6 % of the gate bits came out stale (tools/bwd_debug.py).
template <int L0, int L1>
__device__ __forceinline__ void put_lanes(unsigned& v0, unsigned& v1, unsigned x00, unsigned x01, unsigned x10, unsigned x11) {
  asm("s_nop 1\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %1, %3, %6\n\tv_writelane_b32 %0, %4, %7\n\tv_writelane_b32 %1, %5, %7"
      : "+v"(v0), "+v"(v1)
      : "s"(x00), "s"(x01), "s"(x10), "s"(x11), "n"(L0), "n"(L1));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// End of a ray in pass A.  acc[ht][pt][i] is channel ht*32 + 16h + i (storage order of G) of point pt*32 + (lane & 31), gs_t the g_s
// of those two points.  Stores the relu gates twice -- per sample (mrow: [HD/32][K] dwords of this ray, bit j of dword ht = channel
// ht*32 + j; what the scatter pass reads) and per channel (prow: [HD] x 64 bits, bit p = sample p; the A operand of the dW_pe pass) --
// and adds the lane's share of dw_out = sum relu(h) g_s: dw[ht] belongs to channel ht*32 + 16h + (col >> 1), on both lanes of the
// pair.  The per-channel form is free: the compare that opens a gate leaves exactly that mask over the wave's samples in SGPRs.
template <int HD>
__device__ __forceinline__ void gates_and_dwout(const f32x16 (&acc)[HD / 32][2], const float (&gs_t)[2], unsigned* __restrict__ mrow,
                                                uint2* __restrict__ prow, int K, int lane, float (&dw)[HD / 32]) {
  constexpr int HT = HD / 32;
  const int col = lane & 31;
  unsigned pm0 = 0, pm1 = 0;   // lane = channel: gates of samples 0-31 / 32-63
  static_for<0, HT>([&](auto htc) {
    constexpr int ht = decltype(htc)::value;
    float r[16];
    unsigned x = 0, y = 0;   // gate bits of the lane's 16 channels: point tile 0 / 1
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const float a0 = acc[ht][0][i], a1 = acc[ht][1][i];
      const bool o0 = a0 > 0.0f, o1 = a1 > 0.0f;
      x |= (o0 ? 1u : 0u) << i;
      y |= (o1 ? 1u : 0u) << i;
      // lanes 0-31 of the ballot: channel ht*32 + i of the tile's 32 samples; lanes 32-63: channel ht*32 + 16 + i
      const unsigned long long b0 = __ballot(o0), b1 = __ballot(o1);
      put_lanes<ht * 32 + i, ht * 32 + 16 + i>(pm0, pm1, (unsigned)b0, (unsigned)b1, (unsigned)(b0 >> 32), (unsigned)(b1 >> 32));
      r[i] = __builtin_fmaf(relu1(a1), gs_t[1], relu1(a0) * gs_t[0]);
    });
    // x' = {tile 0 channels 0-15 | tile 1 channels 0-15}, y' = {tile 0 channels 16-31 | tile 1 channels 16-31}: lane l = sample l
    swap32u(x, y);
    if (lane < K) mrow[ht * K + lane] = x | (y << 16);
    dw[ht] += half_transpose_reduce16(r, col);
  });
  if (lane < HD) prow[lane] = make_uint2(pm0, pm1);
}

// Cold path of pass A (see eval_point_exact): some sample's encoding argument leaves the fast sincos range.  The forward evaluated
// this ray with eval_point (fp32-input MFMAs, libm sines); the same here, masks written from inside so that no accumulator
// array crosses the call.
template <int C, int HD>
__device__ __attribute__((noinline)) void rows_exact(const float* lds, const float4* G, const float* w2c, const float* Kc, int H, int W,
                                                     int fs, int code_mode, int inv_z, float inv_dmax, float inv_range, float d_min, float range,
                                                     float freq_factor, int learn_empty, float px, float py, float pz, float gs,
                                                     unsigned* mrow, uint2* prow, int K, float* dw_out /* [HD/32], per lane */) {
  using L = Lds<C, HD, 0, true>;
  constexpr int HT = HD / 32;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const Cam enc = load_cam(w2c, Kc);
  const Proj pe = code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
  const Taps tp = make_taps(pe.x, pe.y, H, W, fs);
  float v3[3];
  v3[0] = pe.x, v3[1] = pe.y;
  v3[2] = depth_code(pe, code_mode == 1, inv_z != 0, inv_dmax, inv_range, d_min, range);
  const bool use_empty = (learn_empty != 0) & pe.invalid;
  f32x16 acc[HT][2];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) acc[ht][pt] = zero_acc();
  int o[2][4];
  float wq[2][4];
  bool emp[2];
  unsigned t0, t1;
  bcast_tiles((unsigned)tp.o00, t0, t1), o[0][0] = (int)t0, o[1][0] = (int)t1;
  bcast_tiles((unsigned)tp.o01, t0, t1), o[0][1] = (int)t0, o[1][1] = (int)t1;
  bcast_tiles((unsigned)tp.o10, t0, t1), o[0][2] = (int)t0, o[1][2] = (int)t1;
  bcast_tiles((unsigned)tp.o11, t0, t1), o[0][3] = (int)t0, o[1][3] = (int)t1;
  bcast_tiles(__float_as_uint(tp.w00), t0, t1), wq[0][0] = __uint_as_float(t0), wq[1][0] = __uint_as_float(t1);
  bcast_tiles(__float_as_uint(tp.w01), t0, t1), wq[0][1] = __uint_as_float(t0), wq[1][1] = __uint_as_float(t1);
  bcast_tiles(__float_as_uint(tp.w10), t0, t1), wq[0][2] = __uint_as_float(t0), wq[1][2] = __uint_as_float(t1);
  bcast_tiles(__float_as_uint(tp.w11), t0, t1), wq[0][3] = __uint_as_float(t0), wq[1][3] = __uint_as_float(t1);
  bcast_tiles(use_empty ? 1u : 0u, t0, t1), emp[0] = t0 != 0, emp[1] = t1 != 0;
  {
    GBuf ga, gb;
    gload<HD>(ga, G, o[0], 0, 4 * h);
    gather_seq<HD, 0>(acc, ga, gb, G, o, wq, h);
  }
  if (learn_empty && __any(use_empty)) apply_empty<HD>(acc, emp, lds + L::EMPTY, h);
  const float* wl = lds + L::W_IN + h * HD + (lane & 31);
  kstep<HD>(acc, wl, 0, v3[0], v3[1]);
  kstep<HD>(acc, wl + 2 * HD, 0, v3[2], 1.0f);
  wl += 4 * HD;
  float ff = freq_factor;
#pragma unroll 1
  for (int oct = 0; oct < kNumFreqs; ++oct) {
    float sc[6];
    pe_octave(sc, v3, ff);
    kstep<HD>(acc, wl, 0, sc[0], sc[1]);
    kstep<HD>(acc, wl + 2 * HD, 0, sc[2], sc[3]);
    kstep<HD>(acc, wl + 4 * HD, 0, sc[4], sc[5]);
    wl += 6 * HD;
    ff = ff * 2.0f;
  }
  float gs_t[2];
  bcast_tiles(__float_as_uint(gs), t0, t1);
  gs_t[0] = __uint_as_float(t0), gs_t[1] = __uint_as_float(t1);
  float dw[HT];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) dw[ht] = 0.0f;
  gates_and_dwout<HD>(acc, gs_t, mrow, prow, K, lane, dw);
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) dw_out[ht] = dw[ht];
}

// ---------------------------------------------------------------------------------------------------------------
// pass A
// ---------------------------------------------------------------------------------------------------------------
template <int C, int HD, int NVMAX>
__global__ __launch_bounds__(256, 2) void rows_kernel(const BwdParams bp) {
  const FwdParams& p = bp.f;
  using L = Lds<C, HD, 0, true>;
  using LH = LdsH<C, HD, 0>;
  constexpr int HT = HD / 32;
  constexpr int NS = 4 * HT;
  __shared__ __attribute__((aligned(16))) float lds[L::TOTAL + LH::TOTAL + 4];
  float* const lh = lds + ((L::TOTAL + 3) & ~3);
  stage_weights<C, HD, 0, true>(lds, p.mlp, p.empty_feature);
  __syncthreads();
  stage_weights_h<C, HD, 0>(lh, lds + L::EMPTY, p.mlp);
  __syncthreads();
  const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE])));
  const float inv_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE + 1])));

  const int lane = threadIdx.x & 63;
  const int h0 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef BTS_GATHER_LDS
  extern __shared__ __attribute__((aligned(128))) char gather_lds[];   // per wave: ring of 3 x 4 KB + 768 B tap table (render_kernel_p)
  GatherLds gl;
  {
    char* base = gather_lds + wave * kGatherLdsPerWave;
    gl.ring = base;
    gl.ring_m0 = (unsigned)(unsigned long)base;
    gl.tab = reinterpret_cast<unsigned*>(base + 3 * 4096);
    gl.m = lane >> 3;
    gl.piece16 = 16u * (unsigned)(((lane & 7) + (lane >> 4)) & 7);
    gl.piece16x = gl.piece16 ^ 64u;
    const int col = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) gl.rd[q] = (unsigned)(col * 128 + ((4 * h0 + q - (col >> 1)) & 7) * 16);
  }
#endif
  const int nwg = gridDim.x;  // multiple of 8
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int wg_per_xcd = nwg >> 3;
  const int xcd = wg / wg_per_xcd;
  const int lw = (wg - xcd * wg_per_xcd) * 4 + wave;
  const int waves_per_xcd = wg_per_xcd * 4;
  // chunk-interleaved ray distribution over the XCDs, as render_kernel_p (one group = one ray here)
  const int CHL = p.chunk_log2;
  const int n_groups = (int)p.groups;   // 32-bit ray indices (render_bwd_impl checks), 64 bits only for element offsets
  const int n_chunks = (n_groups + (1 << CHL) - 1) >> CHL;
  auto group_of = [&](int idx) -> int {
    const int c = ((idx >> CHL) << 3) + xcd;
    const int gg = (c << CHL) + (idx & ((1 << CHL) - 1));
    return (c < n_chunks && gg < n_groups) ? gg : -1;
  };
  const int Bp = p.Bp, K = p.K;
  const int k = lane;
  const bool valid = k < K;
  const int kk = valid ? k : K - 1;
  const bool last = k == K - 1;

  float dw_acc[HT], db_acc = 0.0f;   // this lane's share of dw_out (see gates_and_dwout) and of db_out
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) dw_acc[ht] = 0.0f;
  int sample_end = Bp;
  int sample = 0;
  int idx = lw;
  int g = group_of(idx);
  float z_pre = 0.0f, zn_pre = 0.0f, s_pre = 0.0f, t_pre = 0.0f;
  if (g >= 0) {
    const long pk = (long)g * K + kk;
    z_pre = p.z_samp[pk], zn_pre = p.z_samp[(long)g * K + min(kk + 1, K - 1)];
    s_pre = p.sigma_raw[pk], t_pre = p.trans[pk];
  }
  const int nv3 = p.nv * 3;
  float rec = g >= 0 ? fetch_ray_record(kernarg_view<BwdParams>(), (long)g, nv3, lane) : 0.0f;   // bts_bwd.h: the ray's scalars, one iteration ahead
#ifdef BTS_TICKS
  // sections: 0 head (ray record, camera, per-sample loads issued, geometry, taps, tile broadcast, first gather blocks out)
  //           1 upstream weight gradient + compositing gradient (waits for the per-sample loads)   2 forward pipeline (gather, encoding, lin_in)
  //           3 gate masks + dw_out
  unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
  const unsigned long long t_begin = t_last;
  unsigned n_iter = 0;
#endif

  for (; g >= 0; idx += waves_per_xcd, g = group_of(idx)) {
    auto qb = kernarg_view<BwdParams>();   // this iteration's parameters, re-read where they are used (bts_common.h: kernarg_view)
    asm volatile("" : "+s"(qb));
    const int H = qb->f.H, W = qb->f.W, nv = qb->f.nv, fs = qb->f.fs;
    const long ray = g;
    while (g >= sample_end) ++sample, sample_end += Bp;
    const Cam enc = load_cam(qb->f.w2c_enc + sample * 16, qb->f.K_enc + sample * 9);
    const float4* __restrict__ G = reinterpret_cast<const float4*>(qb->f.proj) + (long)sample * (H >> fs) * (W >> fs) * (HD / 4);
    const RayIn<NVMAX * 3> rin = unpack_ray_record<NVMAX * 3>(qb, rec, nv3);
    const float ox = rin.o[0], oy = rin.o[1], oz = rin.o[2], dx = rin.d[0], dy = rin.d[1], dz = rin.d[2];
    const float z = z_pre, z_nx = zn_pre, s_raw = s_pre, T = t_pre;
    {  // the next ray's per-sample state (and its scalars) land while this ray is evaluated
      const int gn = group_of(idx + waves_per_xcd);
      if (gn >= 0) {
        rec = fetch_ray_record(qb, (long)gn, nv3, lane);
        const long pk = (long)gn * K + kk;
        z_pre = qb->f.z_samp[pk], zn_pre = qb->f.z_samp[(long)gn * K + min(kk + 1, K - 1)];
        s_pre = qb->f.sigma_raw[pk], t_pre = qb->f.trans[pk];
      }
    }
    int lane_off = h0 * HD + (lane & 31), h = h0;
    asm volatile("" : "+v"(lane_off), "+v"(h));   // keep the weight reads inside the persistent loop (see render_kernel_p)
    const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;
    const long pk = ray * K + kk;

    // ---------------- the ray's upstream gradients (one batch of scalar loads: index clamped, the entries beyond nv zeroed by selects -- a
    // condition per entry is a branch, a load and a wait per entry) and the forward's per-sample colours: issued here, used behind the
    // geometry and the first gather blocks (read where they are needed, each is a full memory round trip with the wave idle)
    float g_rgb[NVMAX * 3];
#pragma unroll
    for (int i = 0; i < NVMAX * 3; ++i) g_rgb[i] = rin.g_rgb[i];
    const float g_bkgd = rin.g_bkgd, g_depth = rin.g_depth;
    float cs_v[NVMAX * 3];
    const bool have_cs = qb->f.rgb_samps != nullptr;
#pragma unroll
    for (int i = 0; i < NVMAX * 3; ++i) cs_v[i] = 0.0f;
    if (have_cs) {
      const float* cs = qb->f.rgb_samps + pk * (long)(nv * 3);
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i) cs_v[i] = cs[min(i, nv * 3 - 1)];   // entries beyond nv meet g_rgb = 0
    }
    const float gw_k = qb->g_weights ? qb->g_weights[pk] : 0.0f;
    const float ga_k = qb->g_alphas ? qb->g_alphas[pk] : 0.0f;

    // ---------------- encoder view
    const Proj pe = qb->f.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
    Taps tp = make_taps(pe.x, pe.y, H, W, fs);
    float v3[3];
    v3[0] = pe.x, v3[1] = pe.y;
    v3[2] = depth_code(pe, qb->f.code_mode == 1, qb->f.inv_z != 0, qb->f.inv_dmax, qb->f.inv_range, qb->f.d_min, qb->f.range);
    const bool use_empty = (qb->f.learn_empty != 0) & pe.invalid;
    if (use_empty) tp.w00 = tp.w01 = tp.w10 = tp.w11 = 0.0f;
    tp.w00 *= scale, tp.w01 *= scale, tp.w10 *= scale, tp.w11 *= scale;

    // ---------------- tap offsets / weights of both point tiles on every lane; the first two gather stages go out now: their latency
    // runs under the compositing gradient below (a scan across the wave, exp / log / divide) instead of in front of the MFMA phase
    int o[2][4];
    float wq[2][4];
    bool emp[2];
    {
      unsigned t0, t1;
      bcast_tiles((unsigned)tp.o00, t0, t1), o[0][0] = (int)t0, o[1][0] = (int)t1;
      bcast_tiles((unsigned)tp.o01, t0, t1), o[0][1] = (int)t0, o[1][1] = (int)t1;
      bcast_tiles((unsigned)tp.o10, t0, t1), o[0][2] = (int)t0, o[1][2] = (int)t1;
      bcast_tiles((unsigned)tp.o11, t0, t1), o[0][3] = (int)t0, o[1][3] = (int)t1;
      bcast_tiles(__float_as_uint(tp.w00), t0, t1), wq[0][0] = __uint_as_float(t0), wq[1][0] = __uint_as_float(t1);
      bcast_tiles(__float_as_uint(tp.w01), t0, t1), wq[0][1] = __uint_as_float(t0), wq[1][1] = __uint_as_float(t1);
      bcast_tiles(__float_as_uint(tp.w10), t0, t1), wq[0][2] = __uint_as_float(t0), wq[1][2] = __uint_as_float(t1);
      bcast_tiles(__float_as_uint(tp.w11), t0, t1), wq[0][3] = __uint_as_float(t0), wq[1][3] = __uint_as_float(t1);
      bcast_tiles(use_empty ? 1u : 0u, t0, t1), emp[0] = t0 != 0, emp[1] = t1 != 0;
    }
#ifdef BTS_GATHER_LDS
    unsigned off_next[4];
    GRows rows;
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      gl.tab[lane * 3 + 0] = (unsigned)tp.o00 * (HD * 4u), gl.tab[lane * 3 + 1] = (unsigned)tp.o01 * (HD * 4u), gl.tab[lane * 3 + 2] = (unsigned)tp.o10 * (HD * 4u);   // o11 = o10 + (o01 - o00)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      gl_prologue<HD>(gl, rows, G, off_next);
    }
#else
    GBuf ba, bb;
#ifndef BTS_ROWS_LATE_GATHER
    stage_load<HD, 0>(ba, G, o, h);
    stage_load<HD, 1>(bb, G, o, h);
#endif
#endif

    BW_TICK(0)
    // ---------------- upstream gradient of this sample's weight: g_w = g_depth z + sum_j g_rgb_j . c_kj (+ g_weights_k)
    float g_w = qb->g_depth ? g_depth * z : 0.0f;
    {
      if (qb->f.white_bkgd) g_w += g_bkgd;   // nerf.py:301-304: rgb = sum_k w_k c_k + 1 - sum_k w_k
      g_w += gw_k;
      if (have_cs) {
#pragma unroll
        for (int j = 0; j < NVMAX; ++j)
          g_w += g_rgb[3 * j] * cs_v[3 * j] + g_rgb[3 * j + 1] * cs_v[3 * j + 1] + g_rgb[3 * j + 2] * cs_v[3 * j + 2];
      } else {
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
          if (j < nv) {
            const Cam cj = load_cam(qb->f.w2c_r + ((long)sample * nv + j) * 16, qb->f.K_r + ((long)sample * nv + j) * 9);
            const Proj pc = project<false>(cj, px, py, pz);
            const Taps tc = make_taps(pc.x, pc.y, H, W);
            const float4* img = reinterpret_cast<const float4*>(qb->f.imgs) + ((long)sample * nv + j) * H * W;
            const float4 a = img[tc.o00], b = img[tc.o01], cc = img[tc.o10], d = img[tc.o11];
            const float c0 = ((a.x * tc.w00 + b.x * tc.w01) + cc.x * tc.w10) + d.x * tc.w11;
            const float c1 = ((a.y * tc.w00 + b.y * tc.w01) + cc.y * tc.w10) + d.y * tc.w11;
            const float c2 = ((a.z * tc.w00 + b.z * tc.w01) + cc.z * tc.w10) + d.z * tc.w11;
            g_w += g_rgb[3 * j] * c0 + g_rgb[3 * j + 1] * c1 + g_rgb[3 * j + 2] * c2;
          }
        }
      }
    }

    // ---------------- compositing gradient (nerf.py:283-299):  g_alpha_k = g_w_k T_k - (sum_{m>k} g_w_m w_m) / (1 - alpha_k + 1e-10)
    float g_s = 0.0f;
    {
      float sigma = softplus(s_raw);
      const bool dead = (qb->f.empty_empty != 0) & pe.invalid;   // sigma forced to 0: no gradient
      if (dead) sigma = 0.0f;
      if (qb->f.sigma_noise) sigma += qb->f.sigma_noise[pk];   // nerf.py:279-280: relu(sigma + noise) -- no gradient where the sum is <= 0
      const bool cut = sigma <= 0.0f && qb->f.sigma_noise != nullptr;
      const float delta = last ? 1e10f : (z_nx - z);
      const float ex = transmittance(delta, sigma);
      const bool capped = (qb->f.hard_cap != 0) & last;
      const float alpha = capped ? 1.0f : 1.0f - ex;
#ifdef BTS_ABL_R2   // timing ablation: no scan
      const float S = g_w * alpha;
#else
      const float S = wave_suffix_excl(valid ? g_w * (alpha * T) : 0.0f, lane);
#endif
      float g_alpha = g_w * T - S / (capped ? 1e-10f : ex + 1e-10f);
      g_alpha += ga_k;
      if (!capped && !dead && !cut && valid) g_s = g_alpha * fabsf(delta) * ex * (s_raw > 20.0f ? 1.0f : sigmoidf(s_raw));
      if (valid) qb->gs_ws[pk] = g_s;
    }
    db_acc += g_s;
    BW_TICK(1)
    unsigned* __restrict__ mrow = qb->mask_ws + ray * (long)(HT * K);
    uint2* __restrict__ prow = qb->pmask_ws + ray * (long)HD;

    if (__builtin_expect(__any(pe_needs_exact(v3, qb->f.freq_factor)), 0)) {
      float dwx[HT];   // through memory: no accumulator array may cross the call (it would be demoted to scratch on the hot path)
      rows_exact<C, HD>(lds, G, qb->f.w2c_enc + sample * 16, qb->f.K_enc + sample * 9, H, W, fs, qb->f.code_mode, qb->f.inv_z, qb->f.inv_dmax, qb->f.inv_range, qb->f.d_min,
                        qb->f.range, qb->f.freq_factor, qb->f.learn_empty, px, py, pz, g_s, mrow, prow, K, dwx);
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) dw_acc[ht] += dwx[ht];
      continue;
    }

    // ---------------- h = bilinear(G) + W_pe . PE + b, exactly as render_kernel_p evaluates it (accumulators carry 2^S)
    f32x16 acc[HT][2];
    {
#if defined(BTS_ROWS_LATE_GATHER) && !defined(BTS_GATHER_LDS)   // A/B: gather issued in front of the MFMA phase
      stage_load<HD, 0>(ba, G, o, h);
      stage_load<HD, 1>(bb, G, o, h);
#endif
      f32x16 bias[HT];
      {
        const float* bl = lh + LH::W_RAW + 3 * HD + 4 * h;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(bl + ht * 32 + 8 * j);
            bias[ht][4 * j + 0] = v.x, bias[ht][4 * j + 1] = v.y, bias[ht][4 * j + 2] = v.z, bias[ht][4 * j + 3] = v.w;
          }
      }
      SinCos3 raw;
      pe_direct(raw, v3, qb->f.freq_factor);
      __builtin_amdgcn_sched_barrier(0);
      int lane4 = lane * 4;
      asm volatile("" : "+v"(lane4));
#ifdef BTS_GATHER_LDS
      region_seq_l<HD, 0>(acc, gl, rows, G, wq, off_next, lh + LH::W_F16 + lane4, LH::TERM_STRIDE, raw, v3, qb->f.freq_factor, bias);
      if constexpr (NS > kNumFreqs) {
        gl_consume<HD, 12>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 13>(acc, gl, rows, G, wq, off_next);
        gl_consume<HD, 14>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 15>(acc, gl, rows, G, wq, off_next);
      }
#else
      region_seq<HD, 0>(acc, ba, bb, G, o, wq, h, lh + LH::W_F16 + lane4, LH::TERM_STRIDE, raw, v3, qb->f.freq_factor, bias);
      if constexpr (NS > kNumFreqs) {
        stage_blend<HD, 6>(acc, ba, wq);
        stage_blend<HD, 7>(acc, bb, wq);
      }
#endif
    }
    if (qb->f.learn_empty && __any(use_empty)) {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float ev = lh[LH::EMPTY + ht * 32 + mfma_row(q, 0) + 4 * h];
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) acc[ht][pt][q] += emp[pt] ? ev : 0.0f;
        }
    }

    BW_TICK(2)
    // ---------------- gate masks; dw_out += relu(h) g_s (2^S removed through g_s)
    float gs_t[2];
    {
      unsigned t0, t1;
      bcast_tiles(__float_as_uint(g_s * inv_scale), t0, t1);
      gs_t[0] = __uint_as_float(t0), gs_t[1] = __uint_as_float(t1);
    }
#ifdef BTS_ABL_R1   // timing ablation: no gate masks / dw_out (one store keeps the accumulators alive)
    if (lane < K) mrow[lane] = __float_as_uint(acc[0][0][0] + acc[0][1][5] + acc[HT - 1][0][9] + acc[HT - 1][1][15]);
#else
    gates_and_dwout<HD>(acc, gs_t, mrow, prow, K, lane, dw_acc);
#endif
    BW_TICK(3)
#ifdef BTS_TICKS
    ++n_iter;
#endif
  }
#ifdef BTS_TICKS
  if (bp.ticks && lane == 0 && (long)blockIdx.x * 4 + wave < 4096) {
    unsigned long long* d = bp.ticks + ((long)blockIdx.x * 4 + wave) * 16;
#pragma unroll
    for (int i = 0; i < 13; ++i) d[i] = t_acc[i];
    d[14] = n_iter, d[15] = __builtin_readcyclecounter() - t_begin;
  }
#endif

  // ---------------- dw_out, db_out: wave registers -> work-group LDS -> one atomic per parameter
  if (bp.d_mlp) {
    __shared__ float red[HD + 1];
    for (int i = threadIdx.x; i <= HD; i += blockDim.x) red[i] = 0.0f;
    __syncthreads();
    if ((lane & 1) == 0) {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) atomicAdd(&red[proj_hidden_of_storage(ht * 32 + 16 * h0 + ((lane & 31) >> 1))], dw_acc[ht]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) db_acc += __shfl_xor(db_acc, off, 64);
    if (lane == 0) atomicAdd(&red[HD], db_acc);
    __syncthreads();
    const MlpLayout ml{C + kPeDim, HD, 0};
    for (int i = threadIdx.x; i <= HD; i += blockDim.x) {
      const float v = red[i];
      if (v != 0.0f) flush_add_f32(bp.d_mlp + (i < HD ? ml.w_out() + i : ml.b_out()), v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pass B: dG scatter + d_empty.
// One wave per (patch of 64 rays, 32-channel half of the row).  Neighbouring rays and consecutive samples of a patch land on the same
// few texels of G (the 4 taps x 64 rays of one step cover ~9x9 texels, the footprint drifts by about a pixel per step), so the wave
// keeps a CW x CH texel window of dG rows in LDS (slot = (y mod CH, x mod CW): a texel keeps its slot while the window slides), adds
// the tap contributions there with plain read-modify-write rounds, and only rows LEAVING the window go to global memory as rows of
// float atomics (~5-10 % of the tap updates).  Lanes 0-31 and 32-63 work on DIFFERENT points of the step (points i and i + 32: four
// patch rows apart, their footprints almost never share a texel), so one round serves two points; the taps and the gate mask of
// every point come from a small per-wave LDS table (two broadcast reads per pair); at 20 KB of LDS per wave eight waves fit a CU.
// Pairs whose points do share a window slot (found by comparing the partner's slots, lane = ray) take their rounds one after the
// other.  Points that took the empty feature aim at a dedicated row of the window (one per lane half), so d_empty falls out of the
// same rounds.  A step whose footprint does not fit the window falls back to
// direct row atomics.
// ---------------------------------------------------------------------------------------------------------------
struct ScatterMaskParams {
  FwdParams f;
  const unsigned* mask_ws;   // (n*Bp, HD/32, K) relu gates of lin_in's output, bit j of dword ht = channel ht*32 + j (storage order of G)
  const float* u0_ws;        // ROWS form (bts_bwd_blocks.hip): (n*Bp, K, HD) gradient rows at lin_in's output instead of gates x w_out x g_s
  const float* gs_ws;        // (n*Bp, K)
  float* d_proj;             // or null: only d_empty
  float* d_empty_proj;       // or null
  unsigned char* tiles;      // or null.  (n, tiles_per_img) dirty flags of d_proj (BtsRenderGrads.d_proj_tiles): the byte of every 64-texel
  int tiles_per_img;         // tile that receives a contribution is set to 1
  int tile_tw;               // tile_cols(H >> fs, W >> fs, BtsFieldCfg.tile_blocks)
  int groups_per_sample;
  int w_out_off;             // offset of w_out in the packed parameter vector
#ifdef BTS_TICKS
  unsigned long long* ticks; // diagnostic build: [waves][16] cycles per section of a step (tools/bwd_ticks.py)
#endif
  int nseg, kseg;            // the K steps of a ray group are cut into nseg segments of kseg steps, one wave each (own window, own flush):
                             // a batch of few patches (RE10K: 384, KITTI-Raw: 256) otherwise leaves most of the 1024 SIMDs idle
};

// wave-wide minimum on the DPP network (row_shr 1/2/4/8, row_bcast15, row_bcast31: lane 63 ends up with the result) -- four of these
// per step; as six ds_bpermute round trips each they cost as much as the step's read-modify-write rounds
__device__ __forceinline__ int wave_min_i(int v) {
  auto step = [&](auto ctrl, auto row_mask) {
    const int y = __builtin_amdgcn_update_dpp(v, v, decltype(ctrl)::value, decltype(row_mask)::value, 0xF, false);
    v = min(v, y);
  };
  step(std::integral_constant<int, kDppRowShr + 1>{}, std::integral_constant<int, 0xF>{});
  step(std::integral_constant<int, kDppRowShr + 2>{}, std::integral_constant<int, 0xF>{});
  step(std::integral_constant<int, kDppRowShr + 4>{}, std::integral_constant<int, 0xF>{});
  step(std::integral_constant<int, kDppRowShr + 8>{}, std::integral_constant<int, 0xF>{});
  step(std::integral_constant<int, kDppBcast15>{}, std::integral_constant<int, 0xA>{});
  step(std::integral_constant<int, kDppBcast31>{}, std::integral_constant<int, 0xC>{});
  return __builtin_amdgcn_readlane(v, 63);
}

// ROWS = false: g_h[p][ch] = [gate] w_out[ch] g_s[p] from the gate bits (plain MLP, K <= 64).  ROWS = true: g_h rows from the workspace
// (ResnetBlockFC layers / long rays, bts_bwd_blocks.hip): lane (h, c) keeps channel c of the 32 points of its lane half in registers,
// fetched one step ahead.
template <int HD, bool ROWS = false>
__global__ __launch_bounds__(64) void scatter_kernel(const ScatterMaskParams sp) {
#ifndef BTS_SCATTER_CW
#define BTS_SCATTER_CW 12
#define BTS_SCATTER_CH 12
#endif
  constexpr int CW = BTS_SCATTER_CW, CH = BTS_SCATTER_CH, NSLOT = CW * CH, SCRATCH = NSLOT, EMPTY = NSLOT + 1;   // slot indices; a slot is 32 floats;
                                                                                           // EMPTY, EMPTY + 1: one row per lane half
  constexpr int NW = HD / 32;
  __shared__ __attribute__((aligned(128))) float cache[(NSLOT + 3) * 32];   // aligned: see round()
  __shared__ float4 tab_w[64];   // per point of the step: the four tap weights times g_s
  __shared__ uint2 tab_sm[64];   //                        x: the four slot indices (8 bits each), y: the gate mask of this channel half
  const FwdParams& p = sp.f;
  const int lane = threadIdx.x;
  const int h = lane >> 5, c = lane & 31;
  const unsigned c4 = (unsigned)c * 4u, cbit = 1u << c;
#ifndef BTS_SCATTER_ORDER
#define BTS_SCATTER_ORDER 0
#endif
  // which (unit, segment) a block takes: blocks are dispatched in index order, and the launch lasts until its last block ends
#if BTS_SCATTER_ORDER == 1      // segment-major, the segment of the NEAR samples first
  const int n_units = (int)gridDim.x / sp.nseg;
  const int seg = blockIdx.x / n_units, unit = blockIdx.x - seg * n_units;
#elif BTS_SCATTER_ORDER == 2    // segment-major, the segment of the FAR samples first
  const int n_units = (int)gridDim.x / sp.nseg;
  const int seg_r = blockIdx.x / n_units, unit = blockIdx.x - seg_r * n_units, seg = sp.nseg - 1 - seg_r;
#else
  const int unit = blockIdx.x / sp.nseg, seg = blockIdx.x - unit * sp.nseg;
#endif
  const int grp = unit / NW, wv = unit - grp * NW;
  const int chg = wv * 32 + c;   // this lane's channel of the row
  const int sample = grp / sp.groups_per_sample;
  const int g_in = grp - sample * sp.groups_per_sample;
  const int Bp = p.Bp, K = p.K, fs = p.fs;
  const int H = p.H >> fs, W = p.W >> fs;   // the map in memory (BtsFieldCfg.feat_shift); the taps are computed for p.H x p.W
  const int r_raw = g_in * 64 + lane;
  const bool ray_ok = r_raw < Bp;
  const int r = ray_ok ? r_raw : Bp - 1;
  const long ray = (long)sample * Bp + r;
  const int n_pts = min(64, Bp - g_in * 64);
  const bool scatter = sp.d_proj != nullptr;
  for (int i = lane; i < (NSLOT + 3) * 32; i += 64) cache[i] = 0.0f;
  const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
  const float4 r0 = reinterpret_cast<const float4*>(p.rays)[ray * 2];
  const float4 r1 = reinterpret_cast<const float4*>(p.rays)[ray * 2 + 1];
  const float* zrow = p.z_samp + ray * K;
  const float* gsrow = sp.gs_ws + ray * K;
  const unsigned* mrow = ROWS ? nullptr : sp.mask_ws + (ray * NW + wv) * K;
  float* __restrict__ dG = sp.d_proj + (long)sample * H * W * HD + chg;
  unsigned char* __restrict__ const dirty = sp.tiles ? sp.tiles + (long)sample * sp.tiles_per_img : nullptr;
  const int tw = sp.tile_tw;        // the map's tile geometry (bts_common.h): blocks per row of the 16 x 4 form, 0 = runs of 64 texels
  const float w_out_ch = ROWS ? 0.0f : p.mlp[sp.w_out_off + proj_hidden_of_storage(chg)];
  // ROWS: channel chg of point i of this lane's half at step k is urow[(i K + k) HD]
  const float* urow = ROWS ? sp.u0_ws + ((long)sample * Bp + g_in * 64 + 32 * h) * K * HD + chg : nullptr;
  const int half_pts = min(32, max(0, n_pts - 32 * h));   // points of this lane half that exist
  const long pstride = (long)K * HD;
  int wx = 0, wy = 0;   // window origin (uniform)

  // Evict the slots whose texels lie outside the window at (nwx, nwy): the columns leaving on one side (all rows), then the rows
  // leaving (remaining columns) -- a move of two or three texels touches 30-50 of the 144 slots.  The two lane halves take
  // alternate slots.  Texels outside the image never received anything (taps are clamped): skipped.
  auto flush_slot = [&](int tx, int ty) {
    float* cc = &cache[((int)((unsigned)ty % CH) * CW + (int)((unsigned)tx % CW)) * 32 + c];
    const float v = *cc;
    if (v != 0.0f) {
      atomic_add_f32(dG + ((long)ty * W + tx) * HD, v);
      if (dirty) dirty[tile_of(ty, tx, W, tw)] = 1;   // (same byte from every lane that added something: one request)
      *cc = 0.0f;
    }
  };
  auto flush_rect = [&](int tx0, int tx1, int ty0, int ty1) {   // texels [tx0, tx1) x [ty0, ty1), inside the current window
    tx0 = max(tx0, 0), ty0 = max(ty0, 0), tx1 = min(tx1, W), ty1 = min(ty1, H);
    if (tx1 <= tx0 || ty1 <= ty0) return;
    if (tx1 - tx0 >= ty1 - ty0) {   // the lane halves pair up along the longer side
#pragma unroll 1
      for (int ty = ty0; ty < ty1; ++ty)
#pragma unroll 1
        for (int tx = tx0 + h; tx < tx1 + h; tx += 2)
          if (tx < tx1) flush_slot(tx, ty);
    } else {
#pragma unroll 1
      for (int tx = tx0; tx < tx1; ++tx)
#pragma unroll 1
        for (int ty = ty0 + h; ty < ty1 + h; ty += 2)
          if (ty < ty1) flush_slot(tx, ty);
    }
  };
  auto flush = [&](int nwx, int nwy, bool all) {
    if (all) {
      flush_rect(wx, wx + CW, wy, wy + CH);
      return;
    }
    // columns staying: [sx0, sx1) = old window x new window
    const int sx0 = max(wx, nwx), sx1 = min(wx + CW, nwx + CW);
    if (sx1 <= sx0) {   // disjoint in x: everything leaves
      flush_rect(wx, wx + CW, wy, wy + CH);
      return;
    }
    flush_rect(wx, sx0, wy, wy + CH);            // columns left of the new window
    flush_rect(sx1, wx + CW, wy, wy + CH);       // ... right of it
    const int sy0 = max(wy, nwy), sy1 = min(wy + CH, nwy + CH);
    if (sy1 <= sy0) {
      flush_rect(sx0, sx1, wy, wy + CH);
      return;
    }
    flush_rect(sx0, sx1, wy, sy0);               // rows above
    flush_rect(sx0, sx1, sy1, wy + CH);          // rows below
  };

  // one read-modify-write round: this lane's point takes its four taps from the table.  Slots start at multiples of 128 bytes:
  // the lane's channel is or-ed into the address
  auto round = [&](int pnt, float row_v) {
    const float4 ww = tab_w[pnt];
    const uint2 sm = tab_sm[pnt];
    const float gv = ROWS ? row_v : ((sm.y & cbit) ? w_out_ch : 0.0f);   // g_h = [gate] w_out g_s, g_s rides in the tap weights
    char* const cb = reinterpret_cast<char*>(cache);
    float* c00 = reinterpret_cast<float*>(cb + (((sm.x & 0xFFu) << 7) | c4));
    float* c01 = reinterpret_cast<float*>(cb + ((((sm.x >> 8) & 0xFFu) << 7) | c4));
    float* c10 = reinterpret_cast<float*>(cb + ((((sm.x >> 16) & 0xFFu) << 7) | c4));
    float* c11 = reinterpret_cast<float*>(cb + (((sm.x >> 24) << 7) | c4));
    const float a00 = *c00, a01 = *c01, a10 = *c10, a11 = *c11;
    *c00 = a00 + ww.x * gv;
    *c01 = a01 + ww.y * gv;
    *c10 = a10 + ww.z * gv;
    *c11 = a11 + ww.w * gv;
  };

  const int k_lo = seg * sp.kseg, k_hi = min(K, k_lo + sp.kseg);   // this wave's steps: k_hi - 1 down to k_lo
  if (k_lo >= k_hi) return;
  // lane = ray: a step's depth, g_s and gate word sit K floats apart between neighbouring lanes -- every load instruction touches 64
  // lines for 4 bytes each, the wave's 192 lines do not survive in L1 / L2 from one step to the next (2 048 waves walk theirs at the
  // same time), and the pass fetched 1.3 GB for 50 MB of per-sample inputs (profiles/r03w_train).  With rows that start on 16-byte
  // boundaries (K % 4 == 0) a lane therefore takes FOUR steps per load, one block ahead of its use.
  const bool blk4 = (K & 3) == 0;
  float4 zc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), zq = zc, gc = zc, gq = zc;
  uint4 mc = make_uint4(0u, 0u, 0u, 0u), mq = mc;
  auto load_block = [&](int b) {   // steps 4 b .. 4 b + 3 of this lane's ray
    zq = *reinterpret_cast<const float4*>(zrow + 4 * b);
    gq = ray_ok ? *reinterpret_cast<const float4*>(gsrow + 4 * b) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (!ROWS) mq = ray_ok ? *reinterpret_cast<const uint4*>(mrow + 4 * b) : make_uint4(0u, 0u, 0u, 0u);
  };
  if (blk4) load_block((k_hi - 1) >> 2);
  float z_n = blk4 ? 0.0f : zrow[k_hi - 1], gs_n = (ray_ok && !blk4) ? gsrow[k_hi - 1] : 0.0f;
  unsigned m_n = (!ROWS && ray_ok && !blk4) ? mrow[k_hi - 1] : 0u;
  constexpr int NROW = ROWS ? 32 : 1;
  float cur[NROW], nxt[NROW];
  if constexpr (ROWS) {
#pragma unroll
    for (int i = 0; i < 32; ++i) nxt[i] = i < half_pts ? urow[i * pstride + (long)(k_hi - 1) * HD] : 0.0f;
  }
#ifdef BTS_TICKS
  // sections: 0 step inputs + geometry + taps   1 footprint (four wave minima), window move + flushes   2 slots, conflict test, table
  //           3 read-modify-write rounds (+ the direct-atomics fallback)   12 set-up in front of the first step   13 final flush
  unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
  const unsigned long long t_begin = clock64();
  unsigned n_iter = 0;
  BW_TICK(12)
#endif
  for (int k = k_hi - 1; k >= k_lo; --k) {
    float z = z_n, gs = gs_n;
    unsigned gate = m_n;
    {  // the next step's inputs
      const int kn = max(k - 1, k_lo);
      if (blk4) {
        if ((k & 3) == 3 || k == k_hi - 1) {   // a new block starts: the one fetched a block ago becomes current, the next goes out
          zc = zq, gc = gq, mc = mq;
          const int bn = (k >> 2) - 1;
          if (4 * bn + 3 >= k_lo) load_block(bn);
        }
        const int j = k & 3;   // uniform
        z = j == 0 ? zc.x : j == 1 ? zc.y : j == 2 ? zc.z : zc.w;
        gs = j == 0 ? gc.x : j == 1 ? gc.y : j == 2 ? gc.z : gc.w;
        gate = j == 0 ? mc.x : j == 1 ? mc.y : j == 2 ? mc.z : mc.w;
      } else {
        z_n = zrow[kn], gs_n = ray_ok ? gsrow[kn] : 0.0f;
        if constexpr (!ROWS) m_n = ray_ok ? mrow[kn] : 0u;
      }
      if constexpr (ROWS) {
#pragma unroll
        for (int i = 0; i < 32; ++i) cur[i] = nxt[i];
#pragma unroll
        for (int i = 0; i < 32; ++i) nxt[i] = (i < half_pts && k > k_lo) ? urow[i * pstride + (long)kn * HD] : 0.0f;
      }
    }
    if (__all(gs == 0.0f)) continue;   // e.g. the capped last sample of every ray: nothing to add
    const Proj pe = project<false>(enc, r0.x + z * r0.w, r0.y + z * r1.x, r0.z + z * r1.y);
    int x0, y0, x1, y1;
    Taps tp = make_taps_xy(pe.x, pe.y, p.H, p.W, x0, y0, x1, y1, fs);
    // taps that name the same texel become one: at the far border the second one's weight is exactly 0 (adding it changes nothing),
    // on a down-scaled map (fs > 0) both count
    if (x1 == x0) tp.w00 += tp.w01, tp.w10 += tp.w11, tp.w01 = tp.w11 = 0.0f;
    if (y1 == y0) tp.w00 += tp.w10, tp.w01 += tp.w11, tp.w10 = tp.w11 = 0.0f;
    const bool use_empty = (p.learn_empty != 0) & pe.invalid;
    if constexpr (!ROWS) tp.w00 *= gs, tp.w01 *= gs, tp.w10 *= gs, tp.w11 *= gs;   // ROWS: g_s is part of the rows
    BW_TICK(0)
    bool fits = false;
    if (scatter) {
      const int mnx = wave_min_i(x0), mxx = -wave_min_i(-x1), mny = wave_min_i(y0), mxy = -wave_min_i(-y1);
      fits = (mxx - mnx < CW) && (mxy - mny < CH);
#ifdef BTS_ABL_S2   // timing ablation: the window never moves (wrong results)
      if (false)
#endif
      if (fits && (mnx < wx || mxx >= wx + CW || mny < wy || mxy >= wy + CH)) {
        const int nwx = mnx - (CW - (mxx - mnx + 1)) / 2, nwy = mny - (CH - (mxy - mny + 1)) / 2;
        flush(nwx, nwy, false);
        wx = nwx, wy = nwy;
      }
    }
    BW_TICK(1)
    // window slots of the four taps.  A clamped tap (x1 == x0 or y1 == y0 at the far border: weight exactly 0) would alias its
    // neighbour's slot inside one round; it goes to the scratch row.  An empty-feature point sends g_s to its half's EMPTY row.
    const int rya = (int)((unsigned)y0 % CH) * CW, ryb = (int)((unsigned)y1 % CH) * CW;
    const int cxa = (int)((unsigned)x0 % CW), cxb = (int)((unsigned)x1 % CW);
    const bool ddx = x1 != x0, ddy = y1 != y0;
    int s00 = rya + cxa;
    int s01 = ddx ? rya + cxb : SCRATCH;
    int s10 = ddy ? ryb + cxa : SCRATCH;
    int s11 = (ddx && ddy) ? ryb + cxb : SCRATCH;
    // (one EMPTY row per lane half: with views that look past the encoder's frustum most pairs would otherwise share that slot)
    if (use_empty) s00 = EMPTY + h, s01 = s10 = s11 = SCRATCH, tp.w00 = ROWS ? 1.0f : gs, tp.w01 = tp.w10 = tp.w11 = 0.0f;
    if (!fits && !use_empty) s00 = s01 = s10 = s11 = SCRATCH;   // handled with direct atomics below
    // pairs (i, i + 32) whose points share a slot (bit i): the four slot indices of a point are one dword, the partner's dword is
    // compared byte against byte in its four rotations with the zero-byte test.  The scratch row is shared by design: for the
    // comparison it gets a different id in each lane half
    const unsigned packed = (unsigned)s00 | ((unsigned)s01 << 8) | ((unsigned)s10 << 16) | ((unsigned)s11 << 24);
    unsigned amask;
    {
      const unsigned sc = 0xF0u + (unsigned)h;
      const unsigned cmp = (s00 == SCRATCH ? sc : (unsigned)s00) | ((s01 == SCRATCH ? sc : (unsigned)s01) << 8) |
                           ((s10 == SCRATCH ? sc : (unsigned)s10) << 16) | ((s11 == SCRATCH ? sc : (unsigned)s11) << 24);
      unsigned a, b;
      bcast_tiles(cmp, a, b);
      unsigned hit = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned x = a ^ __builtin_rotateright32(b, 8 * r);
        hit |= (x - 0x01010101u) & ~x;
      }
      amask = (unsigned)__ballot((hit & 0x80808080u) != 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    tab_w[lane] = make_float4(tp.w00, tp.w01, tp.w10, tp.w11);
    tab_sm[lane] = make_uint2(packed, gate);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    BW_TICK(2)
    if (amask == 0) {
      // one straight block: the table reads of later pairs may run ahead of the window's read-modify-write rounds
#ifdef BTS_ABL_S1   // timing ablation: no read-modify-write rounds
      if (wx == 0x12345678)
#endif
#pragma unroll
      for (int i = 0; i < 32; ++i) round(i + 32 * h, cur[ROWS ? i : 0]);
    } else {
      // some pair shares a slot (patches hanging over the image border pile up on the clamped border texels): its two points take
      // their rounds one after the other
#pragma unroll 1
      for (int i = 0; i < 32; ++i) {
        // (the register block is never indexed dynamically -- that would demote it to scratch memory: this path re-reads the row)
        const float rv = (ROWS && i < half_pts) ? urow[i * pstride + (long)k * HD] : 0.0f;
        if ((amask >> i) & 1u) {
          if (h == 0) round(i, rv);
          // (convergent barrier: the two masked rounds must not be merged back into one -- to the compiler, lanes are independent)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (h == 1) round(i + 32, rv);
        } else {
          round(i + 32 * h, rv);
        }
      }
    }
    if (scatter && !fits) {
      // rare (a footprint wider than the window: rays nearly through the encoder's centre): every tap a row of L2 atomics
      auto bc_i = [&](int v, int pnt) { return __builtin_amdgcn_readlane(v, pnt); };
      auto bc_f = [&](float v, int pnt) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), pnt)); };
#pragma unroll 1
      for (int i = 0; i < n_pts; ++i) {
        const float gv = ROWS ? sp.u0_ws[(((long)sample * Bp + g_in * 64 + i) * K + k) * HD + chg]
                              : (((unsigned)bc_i((int)gate, i) & cbit) ? w_out_ch : 0.0f);
        const long ya = (long)bc_i(y0, i) * W, yb = (long)bc_i(y1, i) * W;
        const int xa = bc_i(x0, i), xb = bc_i(x1, i);
        const float w00 = bc_f(tp.w00, i), w01 = bc_f(tp.w01, i), w10 = bc_f(tp.w10, i), w11 = bc_f(tp.w11, i);
        const bool em = bc_i(use_empty ? 1 : 0, i) != 0;
        if (gv != 0.0f && h == 0 && !em) {
          atomic_add_f32(dG + (ya + xa) * HD, w00 * gv);
          atomic_add_f32(dG + (ya + xb) * HD, w01 * gv);
          atomic_add_f32(dG + (yb + xa) * HD, w10 * gv);
          atomic_add_f32(dG + (yb + xb) * HD, w11 * gv);
          if (dirty)
            dirty[tile_of(bc_i(y0, i), xa, W, tw)] = 1, dirty[tile_of(bc_i(y0, i), xb, W, tw)] = 1, dirty[tile_of(bc_i(y1, i), xa, W, tw)] = 1,
            dirty[tile_of(bc_i(y1, i), xb, W, tw)] = 1;
        }
      }
    }
    BW_TICK(3)
#ifdef BTS_TICKS
    ++n_iter;
#endif
  }
  if (scatter) flush(0, 0, true);
#ifdef BTS_TICKS
  BW_TICK(13)
  if (sp.ticks && lane == 0 && blockIdx.x < 4096) {
    unsigned long long* d = sp.ticks + kTicksScatterOffset + (long)blockIdx.x * 16;
#pragma unroll
    for (int i = 0; i < 14; ++i) d[i] = t_acc[i];
    d[14] = n_iter, d[15] = clock64() - t_begin;
  }
#endif
  if (sp.d_empty_proj && h == 0) {
    const float v = cache[EMPTY * 32 + c] + cache[(EMPTY + 1) * 32 + c];
    if (v != 0.0f) atomic_add_f32(sp.d_empty_proj + proj_hidden_of_storage(chg), v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pass C: dW_pe (and db_in: the constant-1 row of the encoding) on the bf16 matrix pipe.
//   dW_pe^T[ch][kin] = w_out[ch] * sum_p gate[p][ch] * (g_s pe)[p][kin]
// The A operand is a 0/1 matrix: exact in bf16.  The B operand g_s * pe spans the full fp32 exponent range (g_s does), so it is cut
// into three bf16 pieces by truncation (8 + 8 + 8 significand bits: the pieces add up to the fp32 value EXACTLY); 0/1 times a bf16
// piece is exact and the MFMA accumulates in fp32 -- the result is the fp32-input MFMA's, at 48 wide MFMAs per ray (32 cycles each,
// vector instructions issue underneath) instead of 128 v_mfma_f32_32x32x2_f32 (64 cycles each with the VALU blocked: 0.6 ms of the
// first lane = sample backward).  Per ray: the encoding is recomputed lane = sample with the forward's fast sines, scaled by g_s,
// cut, and laid out [piece][kin][sample] in a per-wave LDS tile (rows padded to 144 bytes: the 16-byte column reads of the B operand
// then spread over all banks); the A operand comes from the per-channel gate masks pass A stored, eight samples = one byte at a
// time through a 256-entry table of bf16 octets.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct DwpeParams {
  FwdParams f;
  const uint2* pmask_ws;   // (n*Bp, HD) per-channel gates over the ray's 64 samples
  const float* gs_ws;      // (n*Bp, K)
  float* d_mlp;
  float* slots;            // kFlushSlots x (40 x HD): the work-groups' partial sums (bts_bwd.h), folded into d_mlp by dwpe_reduce_kernel
  long rays;               // n * Bp
};

struct DwpeLds {
  static constexpr int PE_ROWS = kPeDim + 1;             // 40
  static constexpr int ROW = 144;                        // bytes per [piece][kin] row: 64 samples x 2 + 16 of padding
  static constexpr int PLANE = PE_ROWS * ROW;            // one piece
  static constexpr int WAVE = 3 * PLANE;                 // 17 280 bytes per wave
  static constexpr int LUT = 0;                          // 256 x 16 bytes: bit j of the index -> bf16 1.0 / 0.0 in position j
  static constexpr int TILES = LUT + 256 * 16;
  static constexpr int TOTAL = TILES + 4 * WAVE;         // 73 216 bytes: two work-groups per CU
};

// g_s * e cut into three bf16 pieces by truncation, [piece][kin][sample]; lane = sample
__device__ __forceinline__ void write_planes(char* tile, const float (&e)[DwpeLds::PE_ROWS], float gs, int lane) {
  using L = DwpeLds;
#pragma unroll
  for (int i = 0; i < L::PE_ROWS; ++i) {
    const float v = e[i] * gs;
    const unsigned b1 = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(b1);
    const unsigned b2 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(b2);
    unsigned short* dst = reinterpret_cast<unsigned short*>(tile + i * L::ROW) + lane;
    dst[0] = (unsigned short)(b1 >> 16);
    dst[L::PLANE / 2] = (unsigned short)(b2 >> 16);
    dst[L::PLANE] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
}
// cold path: some encoding argument of the ray leaves the fast sines' range (as pe_octave); out of line, with its own copy of e[]
__device__ __attribute__((noinline)) void write_planes_exact(char* tile, float x, float y, float code, float freq_factor, float gs) {
  const float v3[3] = {x, y, code};
  float e[DwpeLds::PE_ROWS];
  e[0] = x, e[1] = y, e[2] = code, e[3] = 1.0f;
  float ff = freq_factor;
#pragma unroll 1
  for (int oct = 0; oct < kNumFreqs; ++oct) {
    float sc[6];
    pe_octave_exact(sc, v3, ff);
#pragma unroll
    for (int i = 0; i < 6; ++i) e[4 + 6 * oct + i] = sc[i];
    ff = ff * 2.0f;
  }
  write_planes(tile, e, gs, (int)(threadIdx.x & 63));
}

template <int C, int HD>
__global__ __launch_bounds__(256, 2) void dwpe_kernel(const DwpeParams dp) {
  constexpr int HT = HD / 32;
  using L = DwpeLds;
  constexpr int PE_ROWS = L::PE_ROWS;
  constexpr int D_IN = C + kPeDim;
  const FwdParams& p = dp.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const MlpLayout ml{D_IN, HD, 0};
  {
    u32x4* lut = reinterpret_cast<u32x4*>(smem + L::LUT);
    const unsigned b = threadIdx.x;   // 256 threads: one entry each
    u32x4 e;
#pragma unroll
    for (int d = 0; d < 4; ++d) e[d] = ((b >> (2 * d)) & 1u ? 0x3F80u : 0u) | ((b >> (2 * d + 1)) & 1u ? 0x3F800000u : 0u);
    lut[b] = e;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: a scalar ray loop
  const int h = lane >> 5, col = lane & 31;
  char* const tile = smem + L::TILES + wave * L::WAVE;
  const int Bp = p.Bp, K = p.K;
  f32x16 dw[HT][2];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) dw[ht][kt] = zero_acc();
  const int k = lane, kk = min(k, K - 1);
  // byte offsets of this lane's B-operand rows (kin = col and, clamped, 32 + col) and of its column of samples
  const int brow0 = col * L::ROW + 16 * h, brow1 = min(32 + col, PE_ROWS - 1) * L::ROW + 16 * h;
  const long stride = (long)gridDim.x * 4;
  // per-ray inputs of this lane, fetched one ray ahead (two waves per SIMD do not cover a global load's latency)
  float z_n = 0.0f, gs_n = 0.0f;
  uint2 pm_n[HT];
  auto fetch = [&](long ray) {
    z_n = p.z_samp[ray * K + kk];
    gs_n = k < K ? dp.gs_ws[ray * K + k] : 0.0f;   // lanes past K contribute exact zeros
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) pm_n[ht] = dp.pmask_ws[ray * HD + ht * 32 + col];
  };
  if ((long)blockIdx.x * 4 + wave < dp.rays) fetch((long)blockIdx.x * 4 + wave);
  for (long ray = (long)blockIdx.x * 4 + wave; ray < dp.rays; ray += stride) {
    const int sample = (int)(ray / Bp);
    const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
    const cfp rp = as_const(p.rays) + ray * 8;
    const float z = z_n, gs = gs_n;
    uint2 pm[HT];   // gates of channel ht*32 + col over the ray's samples
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) pm[ht] = pm_n[ht];
    if (ray + stride < dp.rays) fetch(ray + stride);
    const float px = rp[0] + z * rp[3], py = rp[1] + z * rp[4], pz = rp[2] + z * rp[5];
    const Proj pe = p.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
    float v3[3];
    v3[0] = pe.x, v3[1] = pe.y;
    v3[2] = depth_code(pe, p.code_mode == 1, p.inv_z != 0, p.inv_dmax, p.inv_range, p.d_min, p.range);
    // ---- the 40 inputs of lin_in's encoding part in kernel order (kernel_to_ref_input: x, y, code, 1, then per octave 3 sines and
    // 3 "cosines"), times g_s, cut into bf16 pieces
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (__builtin_expect(__any(pe_needs_exact(v3, p.freq_factor)), 0)) {
      write_planes_exact(tile, v3[0], v3[1], v3[2], p.freq_factor, gs);
    } else {
      float e[PE_ROWS];
      e[0] = v3[0], e[1] = v3[1], e[2] = v3[2], e[3] = 1.0f;
      float ff = p.freq_factor;
#ifdef BTS_ABL_D3   // timing ablation: no trigonometry
#pragma unroll
      for (int i = 4; i < PE_ROWS; ++i) e[i] = v3[i % 3] * (float)i;
      if (false)
#endif
#pragma unroll
      for (int r = 0; r < kNumFreqs / 2; ++r) {   // octaves 2r (direct) and 2r + 1 (angle doubling), as the forward's regions
        SinCos3 raw, dbl;
        float t[6];
        pe_direct(raw, v3, ff);
        pe_entries(t, raw, v3, ff);
#pragma unroll
        for (int i = 0; i < 6; ++i) e[4 + 12 * r + i] = t[i];
        pe_double(dbl, raw);
        pe_entries(t, dbl, v3, ff * 2.0f);
#pragma unroll
        for (int i = 0; i < 6; ++i) e[10 + 12 * r + i] = t[i];
        ff = ff * 4.0f;
      }
#ifdef BTS_ABL_D1   // timing ablation: no plane writes
      if (gs == 12345.0f)
#endif
      write_planes(tile, e, gs, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- four k-slices of 16 samples: A[i = channel][k] = gate, B[k][j = kin] = piece of g_s pe
#ifdef BTS_ABL_D2   // timing ablation: no MFMA block
    if (__builtin_amdgcn_readfirstlane(__float_as_int(gs)) == 0x12345678)
#endif
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      bf16x8 a[HT];
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const unsigned word = sl < 2 ? pm[ht].x : pm[ht].y;
        const unsigned byte = (word >> (16 * (sl & 1) + 8 * h)) & 0xFFu;   // samples 16 sl + 8 h .. + 7
        a[ht] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + L::LUT + byte * 16));
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tile + t * L::PLANE + brow0 + 32 * sl));
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tile + t * L::PLANE + brow1 + 32 * sl));
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
          dw[ht][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ht], b0, dw[ht][0], 0, 0, 0);
          dw[ht][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ht], b1, dw[ht][1], 0, 0, 0);
        }
      }
    }
  }
  // ---- flush: D[i = channel in tile][j = kin in tile], register q of a lane of half h holds row mfma_row(q, h), column col.
  // Wave registers -> work-group accumulator (the tiles' memory, dead now) -> one global atomic per parameter, times w_out
  __syncthreads();
  float* d_wpe = reinterpret_cast<float*>(smem + L::TILES);   // [kin][channel]
  for (int i = threadIdx.x; i < PE_ROWS * HD; i += blockDim.x) d_wpe[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int chn = ht * 32 + mfma_row(q, h), kin = kt * 32 + col;
        if (kin < PE_ROWS) atomicAdd(&d_wpe[kin * HD + chn], dw[ht][kt][q]);
      }
  __syncthreads();
  float* slot = dp.slots + (blockIdx.x % kFlushSlots) * (PE_ROWS * HD);
  for (int i = threadIdx.x; i < PE_ROWS * HD; i += blockDim.x) {
    const float v = d_wpe[i] * p.mlp[ml.w_out() + proj_hidden_of_storage(i % HD)];
    if (v != 0.0f) flush_add_f32(slot + i, v);
  }
}

// slots [kFlushSlots][kin][stored channel] -> d_mlp (w_in's encoding columns and b_in: at the start of the packed parameters whatever
// the number of blocks)
template <int C, int HD>
__global__ __launch_bounds__(256) void dwpe_reduce_kernel(float* __restrict__ slots, float* __restrict__ d_mlp) {
  constexpr int D_IN = C + kPeDim;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kFlushRows * HD) return;
  float v = 0.0f;
#pragma unroll
  for (int s = 0; s < kFlushSlots; ++s) {
    v += slots[s * (kFlushRows * HD) + i];
    slots[s * (kFlushRows * HD) + i] = 0.0f;    // the slot copies leave as they must be found: a step's next scale needs no fill of its own
  }
  const int kin = i / HD, hid = proj_hidden_of_storage(i % HD);
  const int src = kernel_to_ref_input<C>(kin + C);
  if (v != 0.0f) atomic_add_f32(d_mlp + (src >= 0 ? hid * D_IN + src : HD * D_IN + hid), v);
}

// ---------------------------------------------------------------------------------------------------------------
// pass C, ROWS form (ResnetBlockFC layers / long rays, bts_bwd_blocks.hip):  dW_pe^T[ch][kin] = sum_p u0[p][ch] pe[p][kin], the
// constant-1 row of the encoding gives db_in.  u0 = the gradient at lin_in's output is a general fp32 row here (no 0/1 factor to
// exploit), so the contraction runs as fp32-input MFMAs (exact products, fp32 accumulation): the A operand A[i = channel][k = sample]
// is read STRAIGHT from the workspace -- lane (h, col) loads channel col of sample 2s + h, two 128-byte rows per instruction, no
// transposition -- and the B operand B[k = sample][j = kin] comes from a per-wave LDS tile the encoding is written to lane = sample.
// ---------------------------------------------------------------------------------------------------------------
struct DwpeRowsParams {
  FwdParams f;
  const float* u0_ws;   // (n*Bp, K, HD), channels in the storage order of G
  float* d_mlp;
  float* slots;         // kFlushSlots x (40 x HD), see DwpeParams
  long rays;            // n * Bp
};

constexpr int kPeLd = kPeDim + 2;   // 41: leading dimension of the [sample][kin] tile (odd: conflict-free rows and columns)

// cold path: some encoding argument of the unit leaves the fast sines' range (as pe_octave); out of line, with its own copy of e[]
__device__ __attribute__((noinline)) void write_pe_tile_exact(float* tile, float x, float y, float code, float freq_factor) {
  const float v3[3] = {x, y, code};
  float* row = tile + (threadIdx.x & 63) * kPeLd;
  row[0] = x, row[1] = y, row[2] = code, row[3] = 1.0f;
  float ff = freq_factor;
#pragma unroll 1
  for (int oct = 0; oct < kNumFreqs; ++oct) {
    float sc[6];
    pe_octave_exact(sc, v3, ff);
#pragma unroll
    for (int i = 0; i < 6; ++i) row[4 + 6 * oct + i] = sc[i];
    ff = ff * 2.0f;
  }
}

// (three work-groups per CU at d_hidden 32: the pass is latency-bound -- row loads from HBM, then the trigonometry, then a chain of
// MFMAs -- and 42 KB of LDS per work-group allow it)
template <int C, int HD>
__global__ __launch_bounds__(256, HD == 32 ? 3 : 2) void dwpe_rows_kernel(const DwpeRowsParams dp) {
  constexpr int HT = HD / 32;
  constexpr int PE_ROWS = kPeDim + 1;   // 40
  constexpr int D_IN = C + kPeDim;
  const FwdParams& p = dp.f;
  __shared__ float tiles[4 * 64 * kPeLd];   // per wave [64 samples][41]; reused as the work-group accumulator at the end
  static_assert(4 * 64 * kPeLd >= PE_ROWS * HD, "the flush buffer lives in the tiles");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, col = lane & 31;
  float* const tile = tiles + wave * 64 * kPeLd;
  const int Bp = p.Bp, K = p.K;
  // One unit = 64 CONSECUTIVE samples of one batch element's (ray, k) list -- the rows of u0 are contiguous in exactly that order, so
  // a ray of 48 samples wastes no k-steps (units run across ray boundaries; lane = sample looks its own ray up)
  const int per_sample = Bp * K;                         // samples per batch element (< 2^31: checked by the launcher)
  const int units_per_sample = (per_sample + 63) >> 6;
  const long units = (long)p.n * units_per_sample;
  f32x16 dw[HT][2];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) dw[ht][kt] = zero_acc();
  const long stride = (long)gridDim.x * 4;
  for (long unit = (long)blockIdx.x * 4 + wave; unit < units; unit += stride) {
    const int sample = (int)(unit / units_per_sample);
    const int s0 = (int)(unit - (long)sample * units_per_sample) * 64;     // first sample of the unit inside its batch element
    const int n_valid = min(64, per_sample - s0);                          // wave-uniform
    // ---- this lane's sample: ray and depth.  Issued BEFORE the row loads below: vmcnt retires in order, so the geometry -- which needs
    // only these three -- would otherwise wait for all 32 row loads as well and nothing would run under their latency
    const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
    const int si = min(s0 + lane, per_sample - 1);
    const int r_in = si / K;
    const long ray = (long)sample * Bp + r_in;
    const float4 r0 = reinterpret_cast<const float4*>(p.rays)[ray * 2];
    const float4 r1 = reinterpret_cast<const float4*>(p.rays)[ray * 2 + 1];
    const float z = p.z_samp[(long)sample * per_sample + si];
    __builtin_amdgcn_sched_barrier(0);
    // ---- A operand: channel ht*32 + col of samples s0 + 2 s2 + h, all k-steps up front: they land under the trigonometry
    // (wave-uniform base + 32-bit lane offset; whole units -- all but a batch element's last -- need no per-load guard)
    float a[HT][32];
    {
      const float* ur = dp.u0_ws + ((long)sample * per_sample + s0) * (long)HD;   // uniform
      // (no branch around the loads -- clamped rows, zeroed below: behind a merge of two paths the compiler waits with vmcnt(0), i.e.
      // for every row, before the geometry may start)
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) {
        const unsigned row = (unsigned)min(2 * s2 + h, n_valid - 1);
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#ifdef BTS_ABL_E3   // timing ablation: no row loads
          a[ht][s2] = (float)(s2 + lane);
#else
          a[ht][s2] = ur[row * (unsigned)HD + (unsigned)(ht * 32 + col)];
#endif
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the row loads to their MFMAs)
    // ---- B operand: the 40 inputs of lin_in's encoding part in kernel order (x, y, code, 1, then per octave 3 sines and 3 "cosines")
    const float px = r0.x + z * r0.w, py = r0.y + z * r1.x, pz = r0.z + z * r1.y;
    const Proj pe = p.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
    float v3[3];
    v3[0] = pe.x, v3[1] = pe.y;
    v3[2] = depth_code(pe, p.code_mode == 1, p.inv_z != 0, p.inv_dmax, p.inv_range, p.d_min, p.range);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (__builtin_expect(__any(pe_needs_exact(v3, p.freq_factor)), 0)) {
      write_pe_tile_exact(tile, v3[0], v3[1], v3[2], p.freq_factor);
    } else {
      float* row = tile + lane * kPeLd;
      row[0] = v3[0], row[1] = v3[1], row[2] = v3[2], row[3] = 1.0f;
      float ff = p.freq_factor;
#ifdef BTS_ABL_E2   // timing ablation: no trigonometry
#pragma unroll
      for (int i = 4; i < 40; ++i) row[i] = v3[i % 3] * (float)i;
      if (false)
#endif
#pragma unroll
      for (int r = 0; r < kNumFreqs / 2; ++r) {   // octaves 2r (direct) and 2r + 1 (angle doubling), as the forward's regions
        SinCos3 raw, dbl;
        float t[6];
        pe_direct(raw, v3, ff);
        pe_entries(t, raw, v3, ff);
#pragma unroll
        for (int i = 0; i < 6; ++i) row[4 + 12 * r + i] = t[i];
        pe_double(dbl, raw);
        pe_entries(t, dbl, v3, ff * 2.0f);
#pragma unroll
        for (int i = 0; i < 6; ++i) row[10 + 12 * r + i] = t[i];
        ff = ff * 4.0f;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 32 k-steps of two samples each
#ifdef BTS_ABL_E1   // timing ablation: no MFMA block (one accumulate keeps the operands alive)
    {
      float acc1 = 0.0f;
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) acc1 += a[0][s2] * tile[(2 * s2 + h) * kPeLd + col];
      dw[0][0][0] += acc1;
    }
    if (false)
#endif
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
      const float* brow = tile + (2 * s2 + h) * kPeLd;
      const float b0 = brow[col];
      const float b1 = col < PE_ROWS - 32 ? brow[32 + col] : 0.0f;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const float av = 2 * s2 + h < n_valid ? a[ht][s2] : 0.0f;   // rows past the batch element's end were loads of its last row
        dw[ht][0] = mfma(av, b0, dw[ht][0]);
        dw[ht][1] = mfma(av, b1, dw[ht][1]);
      }
    }
  }
  // ---- flush: D[i = channel in tile (storage order)][j = kin in tile], register q of a lane of half h holds row mfma_row(q, h),
  // column col.  Wave registers -> work-group accumulator (the tiles' memory, dead now) -> one global atomic per parameter
  __syncthreads();
  float* d_wpe = tiles;   // [kin][channel]
  for (int i = threadIdx.x; i < PE_ROWS * HD; i += blockDim.x) d_wpe[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int chn = ht * 32 + mfma_row(q, h), kin = kt * 32 + col;
        if (kin < PE_ROWS) atomicAdd(&d_wpe[kin * HD + chn], dw[ht][kt][q]);
      }
  __syncthreads();
  float* slot = dp.slots + (blockIdx.x % kFlushSlots) * (PE_ROWS * HD);
  for (int i = threadIdx.x; i < PE_ROWS * HD; i += blockDim.x) {
    const float v = d_wpe[i];
    if (v != 0.0f) flush_add_f32(slot + i, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pass B and pass C side by side
// ---------------------------------------------------------------------------------------------------------------
// Pass B (scatter: dG) and pass C (dW_pe) both read what pass A left and write different things.  One after the other on the caller's
// stream, each is bound by latency at its own occupancy -- and the scatter pass ends in a tail of slow blocks (its slowest block lives
// 1.8 x the average one).  -DBTS_PASS_OVERLAP puts pass C on a side queue of the library (fork after pass A, join before the call's
// last kernel: the caller's stream sees one ordered sequence) so that it can fill what pass B leaves idle.  MEASURED SLOWER and NOT the
// default (round 6, profiles/r06o: backward +7 % at exp_kitti_360.yaml, +6 % exp_kitti_raw.yaml, +4 % exp_re10k.yaml): pass C is a
// persistent grid that takes two work-groups' worth of every CU the moment it starts, and the scatter blocks then queue for what is
// left -- the same finding as for whole per-scale chains side by side (profiles/r05i).
struct PassQueue {
  hipStream_t q;
  hipEvent_t fork, join;
  bool ok;
};
PassQueue* pass_queue() {
#ifndef BTS_PASS_OVERLAP
  return nullptr;
#else
  static thread_local PassQueue* per_dev[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!per_dev[dev]) {
    PassQueue* pq = new PassQueue;
    pq->ok = hipStreamCreateWithFlags(&pq->q, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&pq->fork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&pq->join, hipEventDisableTiming) == hipSuccess;
    per_dev[dev] = pq;
  }
  return per_dev[dev]->ok ? per_dev[dev] : nullptr;
#endif
}
// -> the stream pass C goes to (the side queue, forked from s here; s itself if there is none)
hipStream_t pass_fork(PassQueue* pq, hipStream_t s) {
  if (!pq || hipEventRecord(pq->fork, s) != hipSuccess || hipStreamWaitEvent(pq->q, pq->fork, 0) != hipSuccess) return s;
  return pq->q;
}
void pass_join(PassQueue* pq, hipStream_t side, hipStream_t s) {
  if (side == s) return;
  (void)hipEventRecord(pq->join, side);
  (void)hipStreamWaitEvent(s, pq->join, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------------------------
static void scatter_segments(ScatterMaskParams& sp, long units, int K);

template <int C, int HD>
static int launch_rows(const BwdParams& bp, int n, int grid, hipStream_t s) {
  const FwdParams& p = bp.f;
#ifdef BTS_GATHER_LDS
  constexpr int dyn = 4 * kGatherLdsPerWave;
#else
  constexpr int dyn = 0;
#endif
  auto go = [&](auto kern) {
    if (dyn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    kern<<<grid, 256, dyn, s>>>(bp);
  };
  if (p.nv <= 1) go(rows_kernel<C, HD, 1>);
  else if (p.nv <= 2) go(rows_kernel<C, HD, 2>);
  else if (p.nv <= 4) go(rows_kernel<C, HD, 4>);
  else go(rows_kernel<C, HD, 8>);
  hipError_t e = hipGetLastError();
  const bool want_b = bp.d_proj || bp.d_empty_proj, want_c = bp.d_mlp != nullptr;
  PassQueue* pq = (want_b && want_c) ? pass_queue() : nullptr;
  const hipStream_t sc = (e == hipSuccess && pq) ? pass_fork(pq, s) : s;     // pass C's stream
  if (e == hipSuccess && want_c) {
    DwpeParams dp;
    dp.f = p, dp.pmask_ws = bp.pmask_ws, dp.gs_ws = bp.gs_ws, dp.d_mlp = bp.d_mlp, dp.slots = bp.flush_ws, dp.rays = (long)n * p.Bp;
    const long wgs = (dp.rays + 3) / 4;
    auto kern = dwpe_kernel<C, HD>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DwpeLds::TOTAL);
    if (!bp.flush_clean) (void)hipMemsetAsync(bp.flush_ws, 0, sizeof(float) * kFlushSlots * kFlushRows * HD, sc);
    kern<<<(int)(wgs < grid ? wgs : grid), 256, DwpeLds::TOTAL, sc>>>(dp);   // grid = 2 work-groups per CU
    dwpe_reduce_kernel<C, HD><<<(kFlushRows * HD + 255) / 256, 256, 0, sc>>>(bp.flush_ws, bp.d_mlp);
    e = hipGetLastError();
  }
  if (e == hipSuccess && want_b) {
    const MlpLayout ml{C + kPeDim, HD, 0};
    ScatterMaskParams sp;
    sp.f = p, sp.mask_ws = bp.mask_ws, sp.u0_ws = nullptr, sp.gs_ws = bp.gs_ws, sp.d_proj = bp.d_proj, sp.d_empty_proj = bp.d_empty_proj;
    sp.tiles = bp.tiles, sp.tiles_per_img = bp.tiles_per_img, sp.tile_tw = bp.tile_tw;
#ifdef BTS_TICKS
    sp.ticks = bp.ticks;
#endif
    sp.groups_per_sample = (p.Bp + 63) / 64, sp.w_out_off = ml.w_out();
    const long units = (long)n * sp.groups_per_sample * (HD / 32);
    scatter_segments(sp, units, p.K);
    scatter_kernel<HD><<<(int)(units * sp.nseg), 64, 0, s>>>(sp);
    e = hipGetLastError();
  }
  pass_join(pq, sc, s);
  if (e != hipSuccess) {
    set_error("%s: backward kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

// segments of the K steps per ray group: as many waves as the chip holds AT ONCE -- eight per CU, 20 KB of LDS each -- and not one more
// (rounding up gave exp_re10k.yaml's 384 patches six segments = 2 304 waves on 2 048 places: a second round for the last 256, i.e. twice
// the steps on the critical path; five segments = 1 920 waves finish in one), at least 8 steps each (window set-up + final flush)
int device_cu_count();
static void scatter_segments(ScatterMaskParams& sp, long units, int K) {
  long want = 8L * device_cu_count() / units;
  // ... unless the patches alone (nearly) fill the chip: then FOUR segments each, i.e. several rounds of short waves.  A wave's time
  // depends on its patch (pairs that share window slots, footprints that outgrow the window, many window moves): at exp_kitti_360.yaml's
  // 2 048 patch halves the slowest wave lived 1.8 x the average one and the launch lasted as long as it did (tools/bwd_ticks.py,
  // profiles/r04zz/bwd_ticks.txt); with 8 192 waves of 16 steps the places are refilled as waves finish and a slow patch holds one for a
  // quarter of the time (0.955 -> 0.909 ms of backward per step, profiles/r04zz/scatter_oversub.txt; two segments: 0.934).  Between one
  // and two rounds is the bad zone (above).
  if (want < 2) want = 4;
  const long most = K >= 8 ? K / 8 : 1;
  if (want > most) want = most;
  if (want < 1) want = 1;
  sp.kseg = (int)((K + want - 1) / want);
  sp.nseg = (K + sp.kseg - 1) / sp.kseg;
}

// ROWS form of pass B / pass C for bts_bwd_blocks.hip
int launch_scatter_rows(const BwdParams& bp, const float* u0_ws, int HD, int n, hipStream_t s) {
  const FwdParams& p = bp.f;
  ScatterMaskParams sp;
  sp.f = p, sp.mask_ws = nullptr, sp.u0_ws = u0_ws, sp.gs_ws = bp.gs_ws, sp.d_proj = bp.d_proj, sp.d_empty_proj = bp.d_empty_proj;
  sp.tiles = bp.tiles, sp.tiles_per_img = bp.tiles_per_img, sp.tile_tw = bp.tile_tw;
#ifdef BTS_TICKS
  sp.ticks = bp.ticks;
#endif
  sp.groups_per_sample = (p.Bp + 63) / 64, sp.w_out_off = 0;
  const long units = (long)n * sp.groups_per_sample * (HD / 32);
  scatter_segments(sp, units, p.K);
  if (HD == 64) scatter_kernel<64, true><<<(int)(units * sp.nseg), 64, 0, s>>>(sp);
  else if (HD == 32) scatter_kernel<32, true><<<(int)(units * sp.nseg), 64, 0, s>>>(sp);
  else return BTS_E_UNSUPPORTED;
  return launch_status();
}

int launch_dwpe_rows(const FwdParams& p, const float* u0_ws, float* d_mlp, float* flush_ws, int C, int HD, int NB, int n, int grid, hipStream_t s,
                     bool flush_clean) {
  DwpeRowsParams dp;
  dp.f = p, dp.u0_ws = u0_ws, dp.d_mlp = d_mlp, dp.slots = flush_ws, dp.rays = (long)n * p.Bp;
  if ((long)p.Bp * p.K > 0x7FFFFF00L) return BTS_E_UNSUPPORTED;
  dp.f.n = n;
  const long units = (long)n * (((long)p.Bp * p.K + 63) / 64);
  const long wgs = (units + 3) / 4;
  const long cap = HD == 32 ? (long)grid / 2 * 3 : grid;   // grid = 2 work-groups per CU; this kernel fits 3 at d_hidden 32
  const int g = (int)(wgs < cap ? wgs : cap);
  if (!flush_clean) (void)hipMemsetAsync(flush_ws, 0, sizeof(float) * kFlushSlots * kFlushRows * HD, s);
  if (C == 64 && HD == 64) {
    dwpe_rows_kernel<64, 64><<<g, 256, 0, s>>>(dp);
    dwpe_reduce_kernel<64, 64><<<(kFlushRows * 64 + 255) / 256, 256, 0, s>>>(flush_ws, d_mlp);
  } else if (C == 32 && HD == 32) {
    dwpe_rows_kernel<32, 32><<<g, 256, 0, s>>>(dp);
    dwpe_reduce_kernel<32, 32><<<(kFlushRows * 32 + 255) / 256, 256, 0, s>>>(flush_ws, d_mlp);
  } else return BTS_E_UNSUPPORTED;
  (void)NB;
  return launch_status();
}

// bp.gs_ws: (n*Bp, K) floats, bp.mask_ws: (n*Bp, HD/32, K) dwords, bp.pmask_ws: (n*Bp, HD) x 64 bits; p.groups / chunk_log2 / lpr set for one ray per wave iteration
int launch_bwd_rows(const BwdParams& bp, int C, int HD, int n, int grid, hipStream_t s) {
  if (C == 64 && HD == 64) return launch_rows<64, 64>(bp, n, grid, s);
  if (C == 32 && HD == 32) return launch_rows<32, 32>(bp, n, grid, s);
  return BTS_E_UNSUPPORTED;
}

}  // namespace bts
