// Density-field queries on raw points (BTSNet.forward, models_bts.py:266-338) on the render kernel's software pipeline, and the
// occupancy profile built from them (scripts/inference_setup.py:201-229: the bird's-eye "behind the scenes" output).
//
// lane = POINT: a wave takes 64 points per iteration and runs them through exactly the per-sample work of render_kernel_p --
// projection into the encoder view, the gather of G through LDS (global_load_lds_dwordx4), the f16-split lin_in on the matrix
// pipe, the ResnetBlockFC layers, lin_out, softplus, (optionally) the colour taps of the render views -- minus the compositing.
// Round 2 served this entry point with the compact lane = point kernel of round 1 (field_kernel: fp32-input MFMAs that block the VALU,
// register gather, 239 - 256 VGPRs); the 4.19 M-point occupancy grid of inference_setup.py is the reference's namesake output.
//
// Profile mode: the query points form a dense (Y, Z*X) grid, y slowest (get_pts' order, inference_setup.py:169-187); a wave takes one
// (z, x) COLUMN per iteration, lane = y, so that the reference's post-processing -- sigma := 1 where any view flags the point invalid,
// cumulative sum along y, count of levels whose running sum stays <= 8, / Y -- is a wave scan + a ballot in the same kernel and no
// per-point tensor reaches HBM at all.
#define BTS_NO_LAUNCH_GLUE
#undef BTS_GATHER_REGS   // (the register-gather A/B build, variants/libbts_gatherregs.so, concerns the render kernels: the query kernel exist in the LDS-gather form only)
#include "bts_render_kernel.h"

namespace bts {

struct QueryParams {
  FwdParams f;          // field + xyz (n, P, 3), outputs rgb / invalid / q_sigma (any may be NULL in profile mode)
  int cols;             // profile mode: number of (z, x) columns = stride between two y levels in the point list; 0 = plain query
  int col_len;          // profile mode: Y (<= 64)
  float threshold;      // profile mode: the running sum's limit (8 in the reference)
  float* profile;       // profile mode: (n, cols)
};

template <int C, int HD, int NB, int NVMAX>
__global__ __launch_bounds__(256, 2) void query_kernel_p(const QueryParams qp) {
  const FwdParams& p = qp.f;
  using L = Lds<C, HD, NB, true>;
  using LH = LdsH<C, HD, NB>;
  constexpr int HT = HD / 32;
  constexpr int NS = 4 * HT;
  __shared__ __attribute__((aligned(16))) float lds[L::TOTAL + LH::TOTAL + 4];
  float* const lh = lds + ((L::TOTAL + 3) & ~3);
  stage_weights<C, HD, NB, true>(lds, p.mlp, p.empty_feature);
  __syncthreads();
  stage_weights_h<C, HD, NB>(lh, lds + L::EMPTY, p.mlp);
  __syncthreads();
  const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE])));
  const float inv_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE + 1])));

  const int lane = threadIdx.x & 63;
  const int h0 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
  const int nwg = gridDim.x;  // multiple of 8
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int wg_per_xcd = nwg >> 3;
  const int xcd = wg / wg_per_xcd;
  const int lw = (wg - xcd * wg_per_xcd) * 4 + wave;
  const int waves_per_xcd = wg_per_xcd * 4;
  // chunk-interleaved distribution of the point groups over the XCDs, as render_kernel_p: the groups resident on an XCD at any time
  // are neighbours in the point list (grids: neighbours in space), so their texel footprints overlap in its L2
  const int CHL = p.chunk_log2;
  const int n_groups = (int)p.groups;
  const int n_chunks = (n_groups + (1 << CHL) - 1) >> CHL;
  auto group_of = [&](int idx) -> int {
    const int c = ((idx >> CHL) << 3) + xcd;
    const int gg = (c << CHL) + (idx & ((1 << CHL) - 1));
    return (c < n_chunks && gg < n_groups) ? gg : -1;
  };
  const int P = p.Bp;
  const bool prof = qp.cols > 0;
  const int groups_per_sample = prof ? qp.cols : (P + 63) / 64;
  const float b_out = as_const(p.mlp)[MlpLayout{C + kPeDim, HD, NB}.b_out()];
  // index of this lane's point inside its sample, and whether it exists
  auto point_of = [&](int g_in, bool& ok) -> long {
    if (prof) {
      ok = lane < qp.col_len;
      return (long)min(lane, qp.col_len - 1) * qp.cols + g_in;
    }
    const int i = g_in * 64 + lane;
    ok = i < P;
    return ok ? i : P - 1;
  };

  int sample_end = groups_per_sample;
  int sample = 0;
  int idx = lw;
  int g = group_of(idx);
  // the first group's points
  float xp = 0.0f, yp = 0.0f, zp = 0.0f;
  if (g >= 0) {
    int s0 = g / groups_per_sample;
    bool ok;
    const long pi = point_of(g - s0 * groups_per_sample, ok);
    const float* q = p.xyz + ((long)s0 * P + pi) * 3;
    xp = q[0], yp = q[1], zp = q[2];
  }

  for (; g >= 0; idx += waves_per_xcd, g = group_of(idx)) {
    auto qq = kernarg_view<QueryParams>();   // this iteration's parameters, re-read where they are used (bts_common.h: kernarg_view)
    asm volatile("" : "+s"(qq));
    const int H = qq->f.H, W = qq->f.W, nv = qq->f.nv, fs = qq->f.fs;
    while (g >= sample_end) ++sample, sample_end += groups_per_sample;
    const int g_in = g - (sample_end - groups_per_sample);
    bool valid;
    const long pidx = (long)sample * P + point_of(g_in, valid);
    const Cam enc = load_cam(qq->f.w2c_enc + sample * 16, qq->f.K_enc + sample * 9);
    const float4* __restrict__ G = reinterpret_cast<const float4*>(qq->f.proj) + (long)sample * (H >> fs) * (W >> fs) * (HD / 4);
    const float px = xp, py = yp, pz = zp;
    {  // the next group's points land while this group is evaluated
      const int gn = group_of(idx + waves_per_xcd);
      if (gn >= 0) {
        const int sn = gn / groups_per_sample;
        bool ok;
        const long pi = point_of(gn - sn * groups_per_sample, ok);
        const float* q = qq->f.xyz + ((long)sn * P + pi) * 3;
        xp = q[0], yp = q[1], zp = q[2];
      }
    }
    int h = h0;
    asm volatile("" : "+v"(h));   // keep the weight reads inside the persistent loop (see render_kernel_p)

    // ---------------- encoder view: projection, taps, depth code
    const Proj pe = qq->f.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
    Taps tp = make_taps(pe.x, pe.y, H, W, fs);
    float v3[3];
    v3[0] = pe.x, v3[1] = pe.y;
    v3[2] = depth_code(pe, qq->f.code_mode == 1, qq->f.inv_z != 0, qq->f.inv_dmax, qq->f.inv_range, qq->f.d_min, qq->f.range);
    const bool use_empty = (qq->f.learn_empty != 0) & pe.invalid;
    const Taps tp_enc = tp;   // grid_sample's own taps: a render view that IS the encoder view reuses them (FwdParams::enc_view)
    if (use_empty) tp.w00 = tp.w01 = tp.w10 = tp.w11 = 0.0f;
    tp.w00 *= scale, tp.w01 *= scale, tp.w10 *= scale, tp.w11 *= scale;
    float wq[2][4];
    bool emp[2];
    {
      unsigned t0, t1;
      bcast_tiles(__float_as_uint(tp.w00), t0, t1), wq[0][0] = __uint_as_float(t0), wq[1][0] = __uint_as_float(t1);
      bcast_tiles(__float_as_uint(tp.w01), t0, t1), wq[0][1] = __uint_as_float(t0), wq[1][1] = __uint_as_float(t1);
      bcast_tiles(__float_as_uint(tp.w10), t0, t1), wq[0][2] = __uint_as_float(t0), wq[1][2] = __uint_as_float(t1);
      bcast_tiles(__float_as_uint(tp.w11), t0, t1), wq[0][3] = __uint_as_float(t0), wq[1][3] = __uint_as_float(t1);
      bcast_tiles(use_empty ? 1u : 0u, t0, t1), emp[0] = t0 != 0, emp[1] = t1 != 0;
    }

    float s_raw;
    if (__builtin_expect(__any(pe_needs_exact(v3, qq->f.freq_factor)), 0)) {
      s_raw = eval_point_exact<C, HD, NB>(lds, G, qq->f.w2c_enc + sample * 16, qq->f.K_enc + sample * 9, H, W, fs, qq->f.code_mode, qq->f.inv_z, qq->f.inv_dmax,
                                          qq->f.inv_range, qq->f.d_min, qq->f.range, qq->f.freq_factor, qq->f.learn_empty, b_out, px, py, pz);
    } else {
      // ---------------- h = bilinear(G) + W_pe . PE + b (render_kernel_p's pipeline)
      f32x16 acc[HT][2];
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
      pe_direct(raw, v3, qq->f.freq_factor);
      __builtin_amdgcn_sched_barrier(0);
      int lane4 = lane * 4;
      asm volatile("" : "+v"(lane4));
      region_seq_l<HD, 0>(acc, gl, rows, G, wq, off_next, lh + LH::W_F16 + lane4, LH::TERM_STRIDE, raw, v3, qq->f.freq_factor, bias);
      if constexpr (NS > kNumFreqs) {
        gl_consume<HD, 12>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 13>(acc, gl, rows, G, wq, off_next);
        gl_consume<HD, 14>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 15>(acc, gl, rows, G, wq, off_next);
      }
      if (qq->f.learn_empty && __any(use_empty)) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float ev = lh[LH::EMPTY + ht * 32 + mfma_row(q, 0) + 4 * h];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[ht][pt][q] += emp[pt] ? ev : 0.0f;
          }
      }
      // ---------------- ResnetBlockFC layers (resnetfc.py:53-62), f16 split as render_kernel_p
      if constexpr (NB > 0) {
        int lane4b = lane * 4;
        asm volatile("" : "+v"(lane4b));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f32x16 net[1][2];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float bias0 = lh[LH::BIAS + b * 2 * HD + mfma_row(q, 0) + 4 * h];
            net[0][0][q] = bias0, net[0][1][q] = bias0;
          }
          hidden_layer_h(net, acc, lh + LH::W_BLK + (2 * b) * LH::BLK_LAYER_STRIDE + lane4b, LH::BLK_TERM_STRIDE, inv_scale);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float bias1 = lh[LH::BIAS + b * 2 * HD + HD + mfma_row(q, 0) + 4 * h];
            acc[0][0][q] += bias1, acc[0][1][q] += bias1;
          }
          hidden_layer_h(acc, net, lh + LH::W_BLK + (2 * b + 1) * LH::BLK_LAYER_STRIDE + lane4b, LH::BLK_TERM_STRIDE, inv_scale);
        }
      }
      // ---------------- lin_out
      float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float w2 = lds[L::W_OUT + ht * 32 + mfma_row(q, 0) + 4 * h];
          p0 = __builtin_fmaf(relu1(acc[ht][0][q]), w2, p0);
          p1 = __builtin_fmaf(relu1(acc[ht][1][q]), w2, p1);
        }
      swap32(p0, p1);
      s_raw = __builtin_fmaf(p0 + p1, inv_scale, b_out);
    }
    float sigma = softplus(s_raw);
    if (qq->f.empty_empty) sigma = pe.invalid ? 0.0f : sigma;

    // ---------------- colours / per-view invalid flags (models_bts.py:218-264, 333)
    bool any_inv = pe.invalid;
    if (!qq->f.only_density) {
#pragma unroll
      for (int j = 0; j < NVMAX; ++j) {
        if (j < nv) {
          Taps tc = tp_enc;
          bool inv = pe.invalid;
          if (j != qq->f.enc_view) {   // wave-uniform
            const Cam cj = load_cam(qq->f.w2c_r + ((long)sample * nv + j) * 16, qq->f.K_r + ((long)sample * nv + j) * 9);
            const Proj pc = project<false>(cj, px, py, pz);
            inv = pc.invalid | pe.invalid;
            if (qq->f.rgb) tc = make_taps(pc.x, pc.y, H, W);
          }
          any_inv |= inv;
          if (valid && qq->f.invalid) qq->f.invalid[pidx * nv + j] = inv ? 1.0f : 0.0f;
          if (qq->f.rgb) {
            const float4* img = reinterpret_cast<const float4*>(qq->f.imgs) + ((long)sample * nv + j) * H * W;
            const float4 a = img[tc.o00], b = img[tc.o01], cc = img[tc.o10], d = img[tc.o11];
            if (valid) {
              qq->f.rgb[(pidx * nv + j) * 3 + 0] = ((a.x * tc.w00 + b.x * tc.w01) + cc.x * tc.w10) + d.x * tc.w11;
              qq->f.rgb[(pidx * nv + j) * 3 + 1] = ((a.y * tc.w00 + b.y * tc.w01) + cc.y * tc.w10) + d.y * tc.w11;
              qq->f.rgb[(pidx * nv + j) * 3 + 2] = ((a.z * tc.w00 + b.z * tc.w01) + cc.z * tc.w10) + d.z * tc.w11;
            }
          }
        }
      }
    } else if (valid && qq->f.invalid) {
      qq->f.invalid[pidx] = pe.invalid ? 1.0f : 0.0f;
    }
    if (valid && qq->f.q_sigma) qq->f.q_sigma[pidx] = sigma;

    // ---------------- occupancy profile of this column (inference_setup.py:219-228): sigma := 1 where any view flags the point,
    // running sum along y (lane), fraction of levels whose running sum is still <= threshold
    if (prof) {
      const float a = valid ? (any_inv ? 1.0f : sigma) : 0.0f;
      const float run = seg_scan_add(a, 64, lane);
      const unsigned long long under = __ballot(valid && run <= qq->threshold);
      if (lane == 0) qq->profile[(long)sample * qq->cols + g_in] = (float)__popcll(under) / (float)qq->col_len;
    }
  }
}

FwdParams make_params(const BtsFieldCfg* cfg, const BtsFieldTensors* t);
int render_grid(const FwdParams& p);
int render_chunk_log2(int grid, long groups);

template <int C, int HD, int NB>
static int launch_query_nv(const QueryParams& qp, int grid, hipStream_t s) {
  constexpr int dyn = 4 * kGatherLdsPerWave;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    kern<<<grid, 256, dyn, s>>>(qp);
  };
  const int nv = qp.f.only_density ? 0 : qp.f.nv;
  if (nv <= 1) go(query_kernel_p<C, HD, NB, 1>);
  else if (nv <= 2) go(query_kernel_p<C, HD, NB, 2>);
  else if (nv <= 4) go(query_kernel_p<C, HD, NB, 4>);
  else go(query_kernel_p<C, HD, NB, 8>);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

// plain query: cols = 0; profile: cols = columns per sample, col_len = Y, P = Y * cols
int query_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int P, int only_density, float* rgb, float* invalid,
               float* sigma, int cols, int col_len, float threshold, float* profile, hipStream_t s) {
  QueryParams qp;
  qp.f = make_params(cfg, t);
  qp.f.xyz = xyz, qp.f.Bp = P, qp.f.K = 1, qp.f.only_density = only_density;
  qp.f.rgb = rgb, qp.f.invalid = invalid, qp.f.q_sigma = sigma;
  qp.cols = cols, qp.col_len = col_len, qp.threshold = threshold, qp.profile = profile;
  qp.f.lpr = 64;
  qp.f.groups = (long)cfg->n * (cols > 0 ? cols : (P + 63) / 64);
  if (qp.f.groups > 0x7FF00000L) {
    set_error("%s: too many points in one call (%ld groups)", "bts_field_query", qp.f.groups);
    return BTS_E_UNSUPPORTED;
  }
  const int grid = render_grid(qp.f);
  qp.f.chunk_log2 = render_chunk_log2(grid, qp.f.groups);
  const int C = cfg->C, HD = cfg->d_hidden, NB = cfg->n_blocks;
  if (C == 64 && HD == 64 && NB == 0) return launch_query_nv<64, 64, 0>(qp, grid, s);
  if (C == 32 && HD == 32 && NB == 1) return launch_query_nv<32, 32, 1>(qp, grid, s);
  if (C == 32 && HD == 32 && NB == 0) return launch_query_nv<32, 32, 0>(qp, grid, s);
  set_error("%s: unsupported MLP shape C=%ld d_hidden=%ld n_blocks=%ld", "bts_field_query", C, HD, NB);
  return BTS_E_UNSUPPORTED;
}

}  // namespace bts
