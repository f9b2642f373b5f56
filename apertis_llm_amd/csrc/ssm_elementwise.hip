// Memory-bound companions of the scan on token-major [T, Dn] data (gfx950):
//   - depthwise causal conv1d + SiLU   (reference core.py:368-375)
//   - post-scan skip + gate            (reference core.py:395-396)
//   - fp32 -> {fp32,bf16} cast with an optional transposed copy (compute copies of the expert
//     weights: [E,R,C] and [E,C,R], so forward and dgrad both read K-contiguous operands)
// Geometry shared by the first two: a thread owns ONE 4-channel chunk (16 B fp32 / 8 B bf16) and
// walks tokens, so per-channel reductions (dD, dw, dbias) stay in registers; a block covers
// RP = 256/CPR rows at a time (CPR = Dn/4 chunks per row) and leaves [nblk, ...] partials that a
// deterministic column-sum kernel folds.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float4 ld4(const T *p);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t *p) {
  uint2 u = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T *p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t *p, float4 v) {
  typedef __attribute__((ext_vector_type(4))) bf16_t bf4;
  bf4 o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
  *reinterpret_cast<bf4 *>(p) = o;
}

// sigmoid on the hardware exp2 / rcp (1 ulp each, ~3e-7 relative: far inside the 1e-4 parity bar), as scan_gate.hip's;
// expf + an IEEE division were ~30 VALU instructions per value
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_grad(float x) { float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }

struct Geo { int CPR, RP; };
__host__ __device__ inline Geo make_geo(int64_t Dn) {
  Geo g; g.CPR = (int)(Dn / 4); g.RP = g.CPR >= 256 ? 1 : 256 / g.CPR; return g;
}

// ---------------------------------------------------------------- gate
template <typename TY, typename TIO>
__global__ void __launch_bounds__(256)
ssm_gate_fwd_k(const TY *__restrict__ y, int64_t y_rs, const TIO *__restrict__ xc, int64_t xc_rs,
               const TIO *__restrict__ z, int64_t z_rs, const float *__restrict__ D, TIO *__restrict__ out,
               int64_t out_rs, int64_t T, int Dn) {
  const Geo g = make_geo(Dn);
  const int rr = threadIdx.x / g.CPR;
  if (rr >= g.RP) return;
  for (int c = threadIdx.x - rr * g.CPR; c < g.CPR; c += 256) {
    const float4 d4 = ld4<float>(D + c * 4);
    for (int64_t t = (int64_t)blockIdx.x * g.RP + rr; t < T; t += (int64_t)gridDim.x * g.RP) {
      float4 yv = ld4<TY>(y + t * y_rs + c * 4), xv = ld4<TIO>(xc + t * xc_rs + c * 4), zv = ld4<TIO>(z + t * z_rs + c * 4);
      float4 o = make_float4((yv.x + d4.x * xv.x) * siluf_(zv.x), (yv.y + d4.y * xv.y) * siluf_(zv.y),
                             (yv.z + d4.z * xv.z) * siluf_(zv.z), (yv.w + d4.w * xv.w) * siluf_(zv.w));
      st4<TIO>(out + t * out_rs + c * 4, o);
    }
  }
}

template <typename TY, typename TIO>
__global__ void __launch_bounds__(256)
ssm_gate_bwd_k(const TIO *__restrict__ dout, int64_t dout_rs, const TY *__restrict__ y, int64_t y_rs,
               const TIO *__restrict__ xc, int64_t xc_rs, const TIO *__restrict__ z, int64_t z_rs,
               const float *__restrict__ D, TY *__restrict__ dy, int64_t dy_rs, TIO *__restrict__ dxc, int64_t dxc_rs,
               TIO *__restrict__ dz, int64_t dz_rs, float *__restrict__ dD_part, int64_t T, int Dn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *red = reinterpret_cast<float4 *>(smem);  // [RP][CPR]
  const Geo g = make_geo(Dn);
  const int rr = threadIdx.x / g.CPR;
  const bool on = rr < g.RP;
  for (int c0 = 0; c0 < g.CPR; c0 += 256) {
    const int c = c0 + (on ? threadIdx.x - rr * g.CPR : 0);
    float4 acc = make_float4(0, 0, 0, 0);
    if (on && c < g.CPR) {
      const float4 d4 = ld4<float>(D + c * 4);
      for (int64_t t = (int64_t)blockIdx.x * g.RP + rr; t < T; t += (int64_t)gridDim.x * g.RP) {
        float4 go = ld4<TIO>(dout + t * dout_rs + c * 4), yv = ld4<TY>(y + t * y_rs + c * 4);
        float4 xv = ld4<TIO>(xc + t * xc_rs + c * 4), zv = ld4<TIO>(z + t * z_rs + c * 4);
        float gy[4], gx[4], gz[4];
        const float go_[4] = {go.x, go.y, go.z, go.w}, y_[4] = {yv.x, yv.y, yv.z, yv.w};
        const float x_[4] = {xv.x, xv.y, xv.z, xv.w}, z_[4] = {zv.x, zv.y, zv.z, zv.w}, d_[4] = {d4.x, d4.y, d4.z, d4.w};
        float a_[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float sz = siluf_(z_[j]);
          gy[j] = go_[j] * sz;
          gx[j] = gy[j] * d_[j];
          gz[j] = go_[j] * (y_[j] + d_[j] * x_[j]) * silu_grad(z_[j]);
          a_[j] += gy[j] * x_[j];
        }
        acc = make_float4(a_[0], a_[1], a_[2], a_[3]);
        st4<TY>(dy + t * dy_rs + c * 4, make_float4(gy[0], gy[1], gy[2], gy[3]));
        st4<TIO>(dxc + t * dxc_rs + c * 4, make_float4(gx[0], gx[1], gx[2], gx[3]));
        st4<TIO>(dz + t * dz_rs + c * 4, make_float4(gz[0], gz[1], gz[2], gz[3]));
      }
    }
    __syncthreads();
    if (on && c < g.CPR) red[rr * min(g.CPR, 256) + (c - c0)] = acc;
    __syncthreads();
    if (on && rr == 0 && c < g.CPR) {
      float4 s = red[c - c0];
      for (int r = 1; r < g.RP; ++r) {
        float4 v = red[r * min(g.CPR, 256) + (c - c0)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4 *>(dD_part + (int64_t)blockIdx.x * Dn + c * 4) = s;
    }
  }
}

// ---------------------------------------------------------------- depthwise causal conv + SiLU
constexpr int CONV_TT = 16;  // tokens per thread run

template <typename T, int KW>
__global__ void __launch_bounds__(256)
dwconv_silu_fwd_k(const T *__restrict__ x, int64_t x_rs, const float *__restrict__ w, const float *__restrict__ bias,
                  T *__restrict__ out, int64_t out_rs, int64_t B, int64_t L, int Dn) {
  const Geo g = make_geo(Dn);
  const int rr = threadIdx.x / g.CPR;
  if (rr >= g.RP) return;
  const int64_t runs_per_seq = ceil_div64(L, CONV_TT);
  for (int c = threadIdx.x - rr * g.CPR; c < g.CPR; c += 256) {
    float wv[4][KW], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bv[j] = bias[c * 4 + j];
#pragma unroll
      for (int q = 0; q < KW; ++q) wv[j][q] = w[(c * 4 + j) * KW + q];
    }
    for (int64_t run = (int64_t)blockIdx.x * g.RP + rr; run < B * runs_per_seq; run += (int64_t)gridDim.x * g.RP) {
      const int64_t b = run / runs_per_seq, t0 = (run - b * runs_per_seq) * CONV_TT;
      const T *xb = x + b * L * x_rs + c * 4;
      T *ob = out + b * L * out_rs + c * 4;
      float win[KW][4];  // win[q] = x[t-(KW-1)+q]
#pragma unroll
      for (int q = 0; q < KW - 1; ++q) {
        int64_t tt = t0 - (KW - 1) + q;
        float4 v = tt >= 0 ? ld4<T>(xb + tt * x_rs) : make_float4(0, 0, 0, 0);
        win[q + 1][0] = v.x; win[q + 1][1] = v.y; win[q + 1][2] = v.z; win[q + 1][3] = v.w;
      }
      const int64_t t1 = min(t0 + CONV_TT, L);
      for (int64_t t = t0; t < t1; ++t) {
#pragma unroll
        for (int q = 0; q < KW - 1; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) win[q][j] = win[q + 1][j];
        float4 v = ld4<T>(xb + t * x_rs);
        win[KW - 1][0] = v.x; win[KW - 1][1] = v.y; win[KW - 1][2] = v.z; win[KW - 1][3] = v.w;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = bv[j];
#pragma unroll
          for (int q = 0; q < KW; ++q) s += wv[j][q] * win[q][j];
          s = to_f32(from_f32<T>(s));  // conv output is stored in the activation dtype before SiLU
          o[j] = siluf_(s);
        }
        st4<T>(ob + t * out_rs, make_float4(o[0], o[1], o[2], o[3]));
      }
    }
  }
}

// ---- tile form of the forward (round 2).  The run-per-thread kernel above keeps ONE 8-byte load in flight per thread and
// touches a row as 44 separate pieces: 2.3 TB/s.  Here a work-group takes (batch, 64 tokens): the 64 + KW - 1 input rows
// arrive as 16-byte pieces, several per thread in flight, into an LDS tile laid out like the rows; then thread (rg, p)
// owns piece p of rows rg, rg + RG, ... - its EPC channels' taps stay in registers, the KW input pieces of an output come
// from LDS (consecutive lanes = consecutive 16-byte pieces: conflict-free ds_read_b128), and a wave's stores cover whole
// consecutive rows.  silu through the hardware exp2 / rcp (1 ulp each), as in scan_gate.hip.
constexpr int CONV_TILE_T = 64;
template <typename T> struct Pc16;
template <> struct Pc16<float> {
  static constexpr int EPC = 4;
  __device__ static __forceinline__ void unpack(const uint4 &v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  __device__ static __forceinline__ uint4 pack(const float (&f)[4]) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Pc16<bf16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ void unpack(const uint4 &v, float (&f)[8]) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(u[k] << 16); f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u); }
  }
  __device__ static __forceinline__ uint4 pack(const float (&f)[8]) {
    bf16_t e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = (bf16_t)f[k];
    uint4 v;
    __builtin_memcpy(&v, e, 16);
    return v;
  }
};

// global -> LDS: `total` 16-byte pieces of `nrow` rows (row r of the tile = token t_first + r; rows outside [0, L) are zeros)
__device__ __forceinline__ void conv_tile_in(char *lds, const char *g, int64_t rs_bytes, int64_t t_first, int64_t L, int nrow,
                                             int PCS, int tid) {
  const int total = nrow * PCS;
  for (int i0 = tid; i0 < total; i0 += 1024) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      const int r = i / PCS, pc = i - r * PCS;
      const int64_t t = t_first + r;
      v[u] = (i < total && t >= 0 && t < L) ? *reinterpret_cast<const uint4 *>(g + t * rs_bytes + pc * 16) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      if (i < total) *reinterpret_cast<uint4 *>(lds + (size_t)i * 16) = v[u];
    }
  }
}

template <typename T, int KW>
__global__ void __launch_bounds__(256)
dwconv_silu_fwd_tile_k(const T *__restrict__ x, int64_t x_rs, const float *__restrict__ w, const float *__restrict__ bias,
                       T *__restrict__ out, int64_t out_rs, int64_t L, int PCS, int RG, int nchunks) {
  typedef Pc16<T> PC;
  constexpr int EPC = PC::EPC, TT = CONV_TILE_T;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [TT + KW - 1][PCS * 16 B]
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x / nchunks;
  const int64_t t0 = (int64_t)(blockIdx.x - b * nchunks) * TT;
  const int rows = (int)min((int64_t)TT, L - t0);
  const int rg = tid / PCS, p = tid - rg * PCS;
  const bool on = rg < RG;
  float wv[EPC][KW], bv[EPC];
  if (on) {
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      bv[j] = bias[p * EPC + j];
#pragma unroll
      for (int q = 0; q < KW; ++q) wv[j][q] = w[(p * EPC + j) * KW + q];
    }
  }
  conv_tile_in(smem, reinterpret_cast<const char *>(x + b * L * x_rs), x_rs * (int64_t)sizeof(T), t0 - (KW - 1), L, rows + KW - 1,
               PCS, tid);
  __syncthreads();
  if (!on) return;
  const int rowb = PCS * 16;
  char *ob = reinterpret_cast<char *>(out + (b * L + t0) * out_rs) + p * 16;
  for (int r = rg; r < rows; r += RG) {
    float acc[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[j] = bv[j];
#pragma unroll
    for (int q = 0; q < KW; ++q) {
      float f[EPC];
      PC::unpack(*reinterpret_cast<const uint4 *>(smem + (r + q) * rowb + p * 16), f);
#pragma unroll
      for (int j = 0; j < EPC; ++j) acc[j] += wv[j][q] * f[j];
    }
    float o[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      const float s = to_f32(from_f32<T>(acc[j]));  // conv output is stored in the activation dtype before SiLU
      o[j] = s * sigmoidf_(s);
    }
    *reinterpret_cast<uint4 *>(ob + (int64_t)r * out_rs * (int64_t)sizeof(T)) = PC::pack(o);
  }
}

// (A tile form of the backward - x and dout rows in LDS, a thread walking six consecutive rows of one 16-byte piece with the x
// window and the last KW dpre values in registers - was built and measured: 111 us against 102 us for the run-per-thread
// kernel below at B=40, L=4096, Dn=176; 180 VGPRs leave two waves per SIMD for a walk that is all dependent LDS-read -> FMA
// chains.  Removed; the backward stays as it was.)
// backward: dpre[t] = dout[t]*silu'(pre[t]); dx[t] = sum_q w[q]*dpre[t+(KW-1)-q]; dw,db partials
template <typename T, int KW>
__global__ void __launch_bounds__(256)
dwconv_silu_bwd_k(const T *__restrict__ x, int64_t x_rs, const float *__restrict__ w, const float *__restrict__ bias,
                  const T *__restrict__ dout, int64_t dout_rs, const T *__restrict__ dout2, int64_t dout2_rs,
                  T *__restrict__ dx, int64_t dx_rs,
                  float *__restrict__ dw_part, float *__restrict__ db_part, int64_t B, int64_t L, int Dn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *red = reinterpret_cast<float *>(smem);  // [RP][CPR][4*(KW+1)]
  constexpr int NA = 4 * (KW + 1);
  const Geo g = make_geo(Dn);
  const int rr = threadIdx.x / g.CPR;
  const bool on = rr < g.RP;
  const int64_t runs_per_seq = ceil_div64(L, CONV_TT);
  for (int c0 = 0; c0 < g.CPR; c0 += 256) {
    const int c = c0 + (on ? threadIdx.x - rr * g.CPR : 0);
    float accw[4][KW], accb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { accb[j] = 0.f;
#pragma unroll
      for (int q = 0; q < KW; ++q) accw[j][q] = 0.f; }
    if (on && c < g.CPR) {
      float wv[4][KW], bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bv[j] = bias[c * 4 + j];
#pragma unroll
        for (int q = 0; q < KW; ++q) wv[j][q] = w[(c * 4 + j) * KW + q];
      }
      for (int64_t run = (int64_t)blockIdx.x * g.RP + rr; run < B * runs_per_seq; run += (int64_t)gridDim.x * g.RP) {
        const int64_t b = run / runs_per_seq, t0 = (run - b * runs_per_seq) * CONV_TT;
        const T *xb = x + b * L * x_rs + c * 4;
        const T *gb = dout + b * L * dout_rs + c * 4;
        // dout2: a second gradient of the same output (its other consumer's), added here as autograd's add kernel would have
        // added it - the sum rounded to T - instead of in a pass of its own
        const T *gb2 = dout2 ? dout2 + b * L * dout2_rs + c * 4 : nullptr;
        T *dxb = dx + b * L * dx_rs + c * 4;
        const int64_t t1 = min(t0 + CONV_TT, L);
        // sliding x window win[q] = x[t-(KW-1)+q] and dpre history dp[q] = dpre[t-(KW-1)+q];
        // walk t over [t0, t1+KW-1): dpre[t] for t>=t1 belongs to the next run but feeds our dx
        float win[KW][4], dp[KW][4];
#pragma unroll
        for (int q = 0; q < KW; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) { win[q][j] = 0.f; dp[q][j] = 0.f; }
#pragma unroll
        for (int q = 0; q < KW - 1; ++q) {
          int64_t tt = t0 - (KW - 1) + q;
          if (tt >= 0) { float4 v = ld4<T>(xb + tt * x_rs); win[q + 1][0] = v.x; win[q + 1][1] = v.y; win[q + 1][2] = v.z; win[q + 1][3] = v.w; }
        }
        for (int64_t t = t0; t < t1 + (KW - 1); ++t) {
#pragma unroll
          for (int q = 0; q < KW - 1; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) { win[q][j] = win[q + 1][j]; dp[q][j] = dp[q + 1][j]; }
          if (t < L) {
            float4 v = ld4<T>(xb + t * x_rs), gv = ld4<T>(gb + t * dout_rs);
            if (gb2) {
              const float4 g2 = ld4<T>(gb2 + t * dout2_rs);
              gv = make_float4(to_f32(from_f32<T>(gv.x + g2.x)), to_f32(from_f32<T>(gv.y + g2.y)), to_f32(from_f32<T>(gv.z + g2.z)),
                               to_f32(from_f32<T>(gv.w + g2.w)));
            }
            win[KW - 1][0] = v.x; win[KW - 1][1] = v.y; win[KW - 1][2] = v.z; win[KW - 1][3] = v.w;
            const float g_[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float s = bv[j];
#pragma unroll
              for (int q = 0; q < KW; ++q) s += wv[j][q] * win[q][j];
              s = to_f32(from_f32<T>(s));
              float d = g_[j] * silu_grad(s);
              dp[KW - 1][j] = d;
              if (t < t1) {  // own tokens only: parameter gradients are not double counted
                accb[j] += d;
#pragma unroll
                for (int q = 0; q < KW; ++q) accw[j][q] += d * win[q][j];
              }
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { win[KW - 1][j] = 0.f; dp[KW - 1][j] = 0.f; }
          }
          // dx[t'] with t' = t-(KW-1): sum_q w[q]*dpre[t'+(KW-1)-q] = sum_q w[q]*dp[KW-1-q]
          const int64_t tp = t - (KW - 1);
          if (tp >= t0 && tp < t1) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float s = 0.f;
#pragma unroll
              for (int q = 0; q < KW; ++q) s += wv[j][q] * dp[KW - 1 - q][j];
              o[j] = s;
            }
            st4<T>(dxb + tp * dx_rs, make_float4(o[0], o[1], o[2], o[3]));
          }
        }
      }
    }
    const int cw = min(g.CPR, 256);
    __syncthreads();
    if (on && c < g.CPR) {
      float *r = red + ((int64_t)rr * cw + (c - c0)) * NA;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j * (KW + 1) + KW] = accb[j];
#pragma unroll
        for (int q = 0; q < KW; ++q) r[j * (KW + 1) + q] = accw[j][q];
      }
    }
    __syncthreads();
    if (on && rr == 0 && c < g.CPR) {
      for (int j = 0; j < 4; ++j) {
        for (int q = 0; q <= KW; ++q) {
          float s = 0.f;
          for (int r = 0; r < g.RP; ++r) s += red[((int64_t)r * cw + (c - c0)) * NA + j * (KW + 1) + q];
          if (q < KW) dw_part[((int64_t)blockIdx.x * Dn + c * 4 + j) * KW + q] = s;
          else db_part[(int64_t)blockIdx.x * Dn + c * 4 + j] = s;
        }
      }
    }
    __syncthreads();
  }
}

// out[c] = sum_r in[r][c], fixed order (16 row groups, four independent loads in flight)
__global__ void __launch_bounds__(1024)
colsum_rows_k(const float *__restrict__ in, float *__restrict__ out, int64_t rows, int64_t cols) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < cols) {
    int64_t r = seg;
    for (; r + 48 < rows; r += 64) {
      float a0 = in[r * cols + c], a1 = in[(r + 16) * cols + c], a2 = in[(r + 32) * cols + c], a3 = in[(r + 48) * cols + c];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; r < rows; r += 16) s += in[r * cols + c];
  }
  part[seg][lane] = s;
  __syncthreads();
  if (seg == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][lane];
    out[c] = t;
  }
}

// ---------------------------------------------------------------- residual + dropout
// y = res + keep/(1-p) * x  (the `output_dropout(block_out) + residual` of every sub-block,
// reference core.py:836-837,918-919) in one pass; the mask is the counter hash of (seed, index),
// regenerated by the backward.
template <typename TX, typename TR>
__global__ void __launch_bounds__(256)
dropout_add_fwd_k(const TX *__restrict__ x, const TR *__restrict__ res, TR *__restrict__ y, int64_t n4, float drop_p,
                  uint64_t seed) {
  const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t th = (uint32_t)(drop_p * 65536.f);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n4; v += (int64_t)gridDim.x * blockDim.x) {
    float4 a = ld4<TX>(x + v * 4), r = ld4<TR>(res + v * 4);
    float e[4] = {a.x, a.y, a.z, a.w};
    if (drop_p > 0.f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = drop_keep(seed, 0, v * 4 + j, 0, th) ? e[j] * ks : 0.f;
    }
    st4<TR>(y + v * 4, make_float4(r.x + e[0], r.y + e[1], r.z + e[2], r.w + e[3]));
  }
}

template <typename TG, typename TX>
__global__ void __launch_bounds__(256)
dropout_bwd_k(const TG *__restrict__ g, TX *__restrict__ dx, int64_t n4, float drop_p, uint64_t seed) {
  const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t th = (uint32_t)(drop_p * 65536.f);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n4; v += (int64_t)gridDim.x * blockDim.x) {
    float4 a = ld4<TG>(g + v * 4);
    float e[4] = {a.x, a.y, a.z, a.w};
    if (drop_p > 0.f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = drop_keep(seed, 0, v * 4 + j, 0, th) ? e[j] * ks : 0.f;
    }
    st4<TX>(dx + v * 4, make_float4(e[0], e[1], e[2], e[3]));
  }
}

// ---------------------------------------------------------------- cast (+ transposed copy)
template <typename TO>
__global__ void __launch_bounds__(256)
cast_transpose_k(const float *__restrict__ src, TO *__restrict__ dst, TO *__restrict__ dstT, int R, int C, int ld_dst,
                 int ld_dstT) {
  __shared__ float tile[64][65];
  const int64_t e = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float *s = src + e * (int64_t)R * C;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    float v = (r < R && c < C) ? s[(int64_t)r * C + c] : 0.f;
    tile[i][tx] = v;
    if (dst && r < R && c < ld_dst) dst[e * (int64_t)R * ld_dst + (int64_t)r * ld_dst + c] = from_f32<TO>(v);   // pad columns: 0
  }
  __syncthreads();
  if (dstT)
    for (int i = ty; i < 64; i += 4) {
      int c = c0 + i, r = r0 + tx;
      if (r < ld_dstT && c < C) dstT[e * (int64_t)C * ld_dstT + (int64_t)c * ld_dstT + r] = from_f32<TO>(tile[tx][i]);
    }
}

// The same with four columns per thread (round 3): 16-byte loads, 8-byte (bf16) / 16-byte (fp32) stores of both copies - the
// scalar form above moved its bytes at 2.6 TB/s (2-byte stores, 128 bytes per wave instruction).  Needs C, ld_dst and ld_dstT
// multiples of 4 and 16-byte aligned bases (the launcher checks).
template <typename TO> struct out4;
template <> struct out4<bf16_t> {
  __device__ static __forceinline__ void store(bf16_t *p, float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) bf16_t bf4;
    const bf4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *reinterpret_cast<bf4 *>(p) = v;
  }
};
template <> struct out4<float> {
  __device__ static __forceinline__ void store(float *p, float a, float b, float c, float d) {
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
  }
};
template <typename TO>
__global__ void __launch_bounds__(256)
cast_transpose4_k(const float *__restrict__ src, TO *__restrict__ dst, TO *__restrict__ dstT, int R, int C, int ld_dst,
                  int ld_dstT) {
  __shared__ float tile[64][65];
  const int64_t e = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float *s = src + e * (int64_t)R * C;
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;        // column quad, row (resp. column) within a pass of 16
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = g + 16 * i, r = r0 + rl, c = c0 + q * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && c < C) v = *reinterpret_cast<const float4 *>(s + (int64_t)r * C + c);
    tile[rl][q * 4 + 0] = v.x; tile[rl][q * 4 + 1] = v.y; tile[rl][q * 4 + 2] = v.z; tile[rl][q * 4 + 3] = v.w;
    if (dst && r < R && c < ld_dst) out4<TO>::store(dst + e * (int64_t)R * ld_dst + (int64_t)r * ld_dst + c, v.x, v.y, v.z, v.w);   // pad columns: 0
  }
  __syncthreads();
  if (dstT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cl = g + 16 * i, c = c0 + cl, r = r0 + q * 4;
      if (r < ld_dstT && c < C)
        out4<TO>::store(dstT + e * (int64_t)C * ld_dstT + (int64_t)c * ld_dstT + r, tile[q * 4 + 0][cl], tile[q * 4 + 1][cl],
                        tile[q * 4 + 2][cl], tile[q * 4 + 3][cl]);
    }
  }
}

// The same for MANY weights in one launch (round 4; the training step's compute copies, prepared once per step): a table of
// matrices [R, C] fp32 -> bf16 copy (rows at `rowmap[r]` if given, pitch ld_plain) and bf16 transposed copy (pitch ld_tr), a
// work-group per 64 x 64 tile found by bisection over the entries' first-tile indices.  Pad rows / columns of the
// destinations are never written (the caller zero-fills them once).  The conversion is apertis_cast_transpose's (RNE).
struct WPEntry {
  const float *src; bf16_t *plain; bf16_t *tr; const int32_t *rowmap;
  int32_t R, C, ld_plain, ld_tr, tiles_c, tile0, pad0, pad1;
};
__global__ void __launch_bounds__(256)
weight_prep_k(const WPEntry *__restrict__ tab, int n_entries) {
  __shared__ float tile[64][65];
  const int t = (int)blockIdx.x;
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {                       // the last entry whose first tile is <= t
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= t) lo = mid; else hi = mid - 1;
  }
  const WPEntry en = tab[lo];
  const int lt = t - en.tile0, tr_ = lt / en.tiles_c, r0 = tr_ * 64, c0 = (lt - tr_ * en.tiles_c) * 64;
  const int R = en.R, C = en.C;
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = g + 16 * i, r = r0 + rl, c = c0 + q * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && c < C) {
      v = *reinterpret_cast<const float4 *>(en.src + (int64_t)r * C + c);
      if (en.plain) out4<bf16_t>::store(en.plain + (int64_t)(en.rowmap ? en.rowmap[r] : r) * en.ld_plain + c, v.x, v.y, v.z, v.w);
    }
    tile[rl][q * 4 + 0] = v.x; tile[rl][q * 4 + 1] = v.y; tile[rl][q * 4 + 2] = v.z; tile[rl][q * 4 + 3] = v.w;
  }
  __syncthreads();
  if (en.tr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cl = g + 16 * i, c = c0 + cl, r = r0 + q * 4;
      if (c >= C || r >= R) continue;
      bf16_t *row = en.tr + (int64_t)c * en.ld_tr;
      if (!en.rowmap && r + 3 < R) {
        out4<bf16_t>::store(row + r, tile[q * 4 + 0][cl], tile[q * 4 + 1][cl], tile[q * 4 + 2][cl], tile[q * 4 + 3][cl]);
      } else {
        for (int k = 0; k < 4; ++k)
          if (r + k < R) row[en.rowmap ? en.rowmap[r + k] : r + k] = (bf16_t)tile[q * 4 + k][cl];
      }
    }
  }
}

int gate_blocks(int64_t T, int64_t Dn) {
  Geo g = make_geo(Dn);
  int64_t nb = ceil_div64(T, (int64_t)g.RP * 8);
  return (int)std::max<int64_t>(1, std::min<int64_t>(nb, 2048));
}
int conv_blocks(int64_t B, int64_t L, int64_t Dn) {
  Geo g = make_geo(Dn);
  int64_t runs = B * ceil_div64(L, CONV_TT);
  return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(runs, g.RP), 4096));
}
bool conv_tile_ok(int64_t B, int64_t L, int64_t Dn, int64_t k, int dtype_io, std::initializer_list<const void *> ptrs,
                  std::initializer_list<int64_t> strides) {
  const int64_t es = dtype_io == APERTIS_BF16 ? 2 : 4, epc = 16 / es;
  if (Dn % epc != 0 || Dn / epc > 256 || B * ceil_div64(L, CONV_TILE_T) >= 0x7fffffffLL || k < 2 || k > 4) return false;
  for (const void *q : ptrs) if (((uintptr_t)q) % 16) return false;
  for (int64_t r : strides) if ((r * es) % 16) return false;
  return true;
}
bool rows_ok(int64_t Dn) { return Dn > 0 && Dn % 4 == 0 && Dn <= 4 * 4096; }

}  // namespace

extern "C" int64_t apertis_ssm_gate_bwd_blocks(int64_t T, int64_t Dn) { return gate_blocks(T, Dn); }
extern "C" int64_t apertis_dwconv_bwd_blocks(int64_t B, int64_t L, int64_t Dn) { return conv_blocks(B, L, Dn); }

#define GATE_TYPES(dy_, dio_, ...)                                                                    \
  do {                                                                                                \
    if ((dy_) == APERTIS_F32 && (dio_) == APERTIS_F32) { typedef float TY; typedef float TIO; __VA_ARGS__; }        \
    else if ((dy_) == APERTIS_F32 && (dio_) == APERTIS_BF16) { typedef float TY; typedef bf16_t TIO; __VA_ARGS__; } \
    else if ((dy_) == APERTIS_BF16 && (dio_) == APERTIS_BF16) { typedef bf16_t TY; typedef bf16_t TIO; __VA_ARGS__; } \
    else return APERTIS_ERR_UNSUPPORTED;                                                              \
  } while (0)

extern "C" int apertis_ssm_gate_fwd(const void *y, int64_t y_rs, const void *xc, int64_t xc_rs, const void *z,
                                    int64_t z_rs, const float *D, void *out, int64_t out_rs, int64_t T, int64_t Dn,
                                    int dtype_y, int dtype_io, void *stream) {
  if (!y || !xc || !z || !D || !out || T < 0) return APERTIS_ERR_ARG;
  if (!rows_ok(Dn) || (y_rs | xc_rs | z_rs | out_rs) % 4) return APERTIS_ERR_UNSUPPORTED;
  if (T == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(gate_blocks(T, Dn)), block(256);
  GATE_TYPES(dtype_y, dtype_io, hipLaunchKernelGGL((ssm_gate_fwd_k<TY, TIO>), grid, block, 0, st, (const TY *)y, y_rs,
                                                   (const TIO *)xc, xc_rs, (const TIO *)z, z_rs, D, (TIO *)out, out_rs, T,
                                                   (int)Dn));
  return apertis_check_launch();
}

extern "C" int apertis_ssm_gate_bwd(const void *dout, int64_t dout_rs, const void *y, int64_t y_rs, const void *xc,
                                    int64_t xc_rs, const void *z, int64_t z_rs, const float *D, void *dy, int64_t dy_rs,
                                    void *dxc, int64_t dxc_rs, void *dz, int64_t dz_rs, float *dD_part, float *dD,
                                    int64_t T, int64_t Dn, int dtype_y, int dtype_io, void *stream) {
  if (!dout || !y || !xc || !z || !D || !dy || !dxc || !dz || !dD_part || !dD || T < 0) return APERTIS_ERR_ARG;
  if (!rows_ok(Dn) || (dout_rs | y_rs | xc_rs | z_rs | dy_rs | dxc_rs | dz_rs) % 4) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = gate_blocks(T, Dn);
  Geo g = make_geo(Dn);
  size_t lds = (size_t)g.RP * std::min(g.CPR, 256) * sizeof(float4);
  dim3 grid(nblk), block(256);
  GATE_TYPES(dtype_y, dtype_io, hipLaunchKernelGGL((ssm_gate_bwd_k<TY, TIO>), grid, block, lds, st, (const TIO *)dout,
                                                   dout_rs, (const TY *)y, y_rs, (const TIO *)xc, xc_rs, (const TIO *)z,
                                                   z_rs, D, (TY *)dy, dy_rs, (TIO *)dxc, dxc_rs, (TIO *)dz, dz_rs, dD_part,
                                                   T, (int)Dn));
  hipLaunchKernelGGL(colsum_rows_k, dim3((unsigned)ceil_div64(Dn, 64)), dim3(1024), 0, st, dD_part, dD, (int64_t)nblk, Dn);
  return apertis_check_launch();
}

#define CONV_DISPATCH(k_, dt_, ...)                                                             \
  do {                                                                                          \
    if ((dt_) == APERTIS_F32) { typedef float T;                                                \
      if ((k_) == 2) { constexpr int KW = 2; __VA_ARGS__; } else if ((k_) == 3) { constexpr int KW = 3; __VA_ARGS__; } \
      else if ((k_) == 4) { constexpr int KW = 4; __VA_ARGS__; } else return APERTIS_ERR_UNSUPPORTED; }  \
    else if ((dt_) == APERTIS_BF16) { typedef bf16_t T;                                         \
      if ((k_) == 2) { constexpr int KW = 2; __VA_ARGS__; } else if ((k_) == 3) { constexpr int KW = 3; __VA_ARGS__; } \
      else if ((k_) == 4) { constexpr int KW = 4; __VA_ARGS__; } else return APERTIS_ERR_UNSUPPORTED; }  \
    else return APERTIS_ERR_ARG;                                                                \
  } while (0)

extern "C" int apertis_dwconv_silu_fwd(const void *x, int64_t x_rs, const float *w, const float *bias, void *out,
                                       int64_t out_rs, int64_t B, int64_t L, int64_t Dn, int64_t k, int dtype_io,
                                       void *stream) {
  if (!x || !w || !bias || !out || B < 0 || L < 0) return APERTIS_ERR_ARG;
  if (!rows_ok(Dn) || (x_rs | out_rs) % 4) return APERTIS_ERR_UNSUPPORTED;
  if (B == 0 || L == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  // tile form when every row is whole 16-byte pieces at 16-byte-aligned addresses (the model's shapes), else run per thread
  if (conv_tile_ok(B, L, Dn, k, dtype_io, {x, out}, {x_rs, out_rs})) {
    const int64_t epc = dtype_io == APERTIS_BF16 ? 8 : 4;
    const int PCS = (int)(Dn / epc), RG = 256 / PCS, nchunks = (int)ceil_div64(L, CONV_TILE_T);
    const size_t lds = (size_t)(CONV_TILE_T + k - 1) * PCS * 16;
    CONV_DISPATCH(k, dtype_io, hipLaunchKernelGGL((dwconv_silu_fwd_tile_k<T, KW>), dim3((unsigned)(B * nchunks)), dim3(256), lds, st,
                                                  (const T *)x, x_rs, w, bias, (T *)out, out_rs, L, PCS, RG, nchunks));
    return apertis_check_launch();
  }
  dim3 grid(conv_blocks(B, L, Dn)), block(256);
  CONV_DISPATCH(k, dtype_io, hipLaunchKernelGGL((dwconv_silu_fwd_k<T, KW>), grid, block, 0, st, (const T *)x, x_rs, w, bias,
                                                (T *)out, out_rs, B, L, (int)Dn));
  return apertis_check_launch();
}

extern "C" int apertis_dwconv_silu_bwd2(const void *x, int64_t x_rs, const float *w, const float *bias, const void *dout,
                                        int64_t dout_rs, const void *dout2, int64_t dout2_rs, void *dx, int64_t dx_rs,
                                        float *dw_part, float *db_part, float *dw, float *db, int64_t B, int64_t L, int64_t Dn,
                                        int64_t k, int dtype_io, void *stream) {
  if (!x || !w || !bias || !dout || !dx || !dw_part || !db_part || !dw || !db || B < 0 || L < 0) return APERTIS_ERR_ARG;
  if (!dout2) dout2_rs = 0;
  if (!rows_ok(Dn) || (x_rs | dout_rs | dout2_rs | dx_rs) % 4 || k < 2 || k > 4) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = conv_blocks(B, L, Dn);
  {
    Geo g = make_geo(Dn);
    size_t lds = (size_t)g.RP * std::min(g.CPR, 256) * 4 * (k + 1) * sizeof(float);
    dim3 grid(nblk), block(256);
    CONV_DISPATCH(k, dtype_io, hipLaunchKernelGGL((dwconv_silu_bwd_k<T, KW>), grid, block, lds, st, (const T *)x, x_rs, w,
                                                  bias, (const T *)dout, dout_rs, (const T *)dout2, dout2_rs, (T *)dx, dx_rs,
                                                  dw_part, db_part, B, L, (int)Dn));
  }
  hipLaunchKernelGGL(colsum_rows_k, dim3((unsigned)ceil_div64(Dn * k, 64)), dim3(1024), 0, st, dw_part, dw, (int64_t)nblk,
                     Dn * k);
  hipLaunchKernelGGL(colsum_rows_k, dim3((unsigned)ceil_div64(Dn, 64)), dim3(1024), 0, st, db_part, db, (int64_t)nblk, Dn);
  return apertis_check_launch();
}

extern "C" int apertis_dwconv_silu_bwd(const void *x, int64_t x_rs, const float *w, const float *bias, const void *dout,
                                       int64_t dout_rs, void *dx, int64_t dx_rs, float *dw_part, float *db_part,
                                       float *dw, float *db, int64_t B, int64_t L, int64_t Dn, int64_t k, int dtype_io,
                                       void *stream) {
  return apertis_dwconv_silu_bwd2(x, x_rs, w, bias, dout, dout_rs, nullptr, 0, dx, dx_rs, dw_part, db_part, dw, db, B, L, Dn, k,
                                  dtype_io, stream);
}

extern "C" int apertis_cast_transpose(const float *src, void *dst, void *dstT, int64_t E, int64_t R, int64_t C,
                                      int64_t ld_dst, int64_t ld_dstT, int dtype_out, void *stream) {
  if (!src || (!dst && !dstT) || E <= 0 || R <= 0 || C <= 0) return APERTIS_ERR_ARG;
  if (ld_dst == 0) ld_dst = C;
  if (ld_dstT == 0) ld_dstT = R;
  // the pad columns are written by the 64-wide tile that holds the last real column
  if (ld_dst < C || ld_dst > ceil_div64(C, 64) * 64 || ld_dstT < R || ld_dstT > ceil_div64(R, 64) * 64) return APERTIS_ERR_ARG;
  if (R > 0x3fffffff || C > 0x3fffffff || E > 65535 || ceil_div64(R, 64) > 65535) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(C, 64), (unsigned)ceil_div64(R, 64), (unsigned)E), block(256);
  const bool vec4 = C % 4 == 0 && ld_dst % 4 == 0 && ld_dstT % 4 == 0 && (R * C) % 4 == 0 && ((uintptr_t)src) % 16 == 0 &&
                    ((uintptr_t)dst) % 16 == 0 && ((uintptr_t)dstT) % 16 == 0;
#define CT_GO(KERN, TO) hipLaunchKernelGGL(KERN<TO>, grid, block, 0, st, src, (TO *)dst, (TO *)dstT, (int)R, (int)C, (int)ld_dst, (int)ld_dstT)
  if (dtype_out == APERTIS_BF16) {
    if (vec4) CT_GO(cast_transpose4_k, bf16_t); else CT_GO(cast_transpose_k, bf16_t);
  } else if (dtype_out == APERTIS_F32) {
    if (vec4) CT_GO(cast_transpose4_k, float); else CT_GO(cast_transpose_k, float);
  } else {
    return APERTIS_ERR_ARG;
  }
#undef CT_GO
  return apertis_check_launch();
}

extern "C" int64_t apertis_weight_prep_entry_bytes(void) { return (int64_t)sizeof(WPEntry); }
extern "C" int apertis_weight_prep(const void *table, int64_t n_entries, int64_t total_tiles, void *stream) {
  if (!table || n_entries <= 0 || total_tiles <= 0 || n_entries > 0x7fffffff || total_tiles > 0x7fffffff) return APERTIS_ERR_ARG;
  hipLaunchKernelGGL(weight_prep_k, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const WPEntry *)table,
                     (int)n_entries);
  return apertis_check_launch();
}

extern "C" int apertis_colsum_f32(const float *in, float *out, int64_t rows, int64_t cols, void *stream) {
  if (!in || !out || rows < 0 || cols <= 0) return APERTIS_ERR_ARG;
  hipLaunchKernelGGL(colsum_rows_k, dim3((unsigned)ceil_div64(cols, 64)), dim3(1024), 0, (hipStream_t)stream, in, out, rows,
                     cols);
  return apertis_check_launch();
}

extern "C" int apertis_dropout_add_fwd(const void *x, const void *res, void *y, int64_t n, float drop_p, uint64_t seed,
                                       int dtype_x, int dtype_res, void *stream) {
  if (!x || !res || !y || n < 0 || drop_p < 0.f || drop_p >= 1.f) return APERTIS_ERR_ARG;
  if (n % 4) return APERTIS_ERR_UNSUPPORTED;
  if (n == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = std::min<int64_t>(ceil_div64(n / 4, 256), 8192);
  dim3 grid((unsigned)nb), block(256);
  if (dtype_x == APERTIS_BF16 && dtype_res == APERTIS_F32)
    hipLaunchKernelGGL((dropout_add_fwd_k<bf16_t, float>), grid, block, 0, st, (const bf16_t *)x, (const float *)res, (float *)y, n / 4, drop_p, seed);
  else if (dtype_x == APERTIS_F32 && dtype_res == APERTIS_F32)
    hipLaunchKernelGGL((dropout_add_fwd_k<float, float>), grid, block, 0, st, (const float *)x, (const float *)res, (float *)y, n / 4, drop_p, seed);
  else if (dtype_x == APERTIS_BF16 && dtype_res == APERTIS_BF16)
    hipLaunchKernelGGL((dropout_add_fwd_k<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t *)x, (const bf16_t *)res, (bf16_t *)y, n / 4, drop_p, seed);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}

extern "C" int apertis_dropout_bwd(const void *g, void *dx, int64_t n, float drop_p, uint64_t seed, int dtype_g,
                                   int dtype_x, void *stream) {
  if (!g || !dx || n < 0 || drop_p < 0.f || drop_p >= 1.f) return APERTIS_ERR_ARG;
  if (n % 4) return APERTIS_ERR_UNSUPPORTED;
  if (n == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = std::min<int64_t>(ceil_div64(n / 4, 256), 8192);
  dim3 grid((unsigned)nb), block(256);
  if (dtype_g == APERTIS_F32 && dtype_x == APERTIS_BF16)
    hipLaunchKernelGGL((dropout_bwd_k<float, bf16_t>), grid, block, 0, st, (const float *)g, (bf16_t *)dx, n / 4, drop_p, seed);
  else if (dtype_g == APERTIS_F32 && dtype_x == APERTIS_F32)
    hipLaunchKernelGGL((dropout_bwd_k<float, float>), grid, block, 0, st, (const float *)g, (float *)dx, n / 4, drop_p, seed);
  else if (dtype_g == APERTIS_BF16 && dtype_x == APERTIS_BF16)
    hipLaunchKernelGGL((dropout_bwd_k<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t *)g, (bf16_t *)dx, n / 4, drop_p, seed);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}
