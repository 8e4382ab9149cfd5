// Bandwidth-bound layer products of the MLP update (gfx950).
//
// An MLP whose input (observations) or head (action mean / value) is narrow has
// four products per minibatch in which one dimension is <= 32:
//   first-layer forward      H1 = tanh(X W0^T + b0)        K = obs_dim
//   head data gradient       dZ = (dout W_head) * (1 - H^2) K = act_dim
//   first-layer weight grad  dW0 = dZ1^T X                  one side = obs_dim
//   head weight grad         dW_head = dout^T H             one side = act_dim
// (reference: the nn.Linear forward/backward of torch/modules/mlp_module.py:62-73
// under torch/algos/vpg.py:250-293).  Each moves one full [rows x hidden]
// activation matrix (33.5 MB at the C3 minibatch) for a few flops per byte, so
// they are HBM/latency bound: an MFMA tile pipeline (load -> LDS -> MFMA -> store,
// one resident wave of workgroups in lock step) exposes every phase.  Here they
// are plain streaming kernels: a thread owns 4 adjacent hidden columns, rows
// stream through registers with several 16-B loads in flight per thread, the
// narrow operand is staged in LDS and read back as same-address broadcasts, and
// the arithmetic is a sequential fp32 FMA chain on the vector ALU.
#include "common.h"
#include <stdlib.h>
#include <hip/hip_ext.h>

#include "prof.h"

namespace {

__device__ __forceinline__ float tanh_fast(float x) {
  return ga_tanh(x);  // common.h
}

// ---------------------------------------------------------------------------
// Y[m, n] = epi(sum_k X[row(m), k] * W(n, k)),  K <= 32, N = 4 * QPR * n_colblk
// ---------------------------------------------------------------------------
struct SkinnyFwdParams {
  const float* X;        // [rows][ldx], K valid floats per row (ldx % 4 == 0)
  int64_t ldx;
  const int32_t* idx;    // optional row gather
  const float* W;        // w_kc: W[n][ldw] (k contiguous) else W[k][ldw]
  int64_t ldw;
  const float* bias;     // epi 0: per-n bias (may be null)
  const float* H;        // epi 1: tanh outputs H[m][ldh]
  int64_t ldh;
  float* Y;
  int64_t ldy;
  int M, N, K;
  int act;               // epi 0: 1 = tanh
  int qpr;               // column quads per row handled by one workgroup (<= 64)
  int rows_per_thread;   // multiple of SK_UNROLL
};

typedef float sk_v2f __attribute__((ext_vector_type(2)));

constexpr int SK_THREADS = 256;
constexpr int SK_UNROLL = 4;

// KV = round4(K) / 4 float4 of the narrow operand per row.  The X rows of the
// workgroup are staged once in LDS (one gathered 16-B load per vector) and read
// back as same-address broadcasts, so the per-row cost is KV ds_read_b128 + 16 KV
// FMAs + one 16-B store per thread.
template <int KV, bool W_KC, int EPI>
__global__ __launch_bounds__(SK_THREADS) void skinny_fwd_kernel(SkinnyFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [rows][4 KV]
  const int tid = threadIdx.x;
  const int q = tid % p.qpr;                 // column quad inside the block
  const int rg = tid / p.qpr;                // row group
  const int n_rg = SK_THREADS / p.qpr;
  const int n0 = (blockIdx.y * p.qpr + q) * 4;
  const bool col_ok = n0 < p.N;              // N % 4 == 0
  const int nc = col_ok ? n0 : 0;
  const int rows_per_block = n_rg * p.rows_per_thread;
  const int row0 = blockIdx.x * rows_per_block;
  const int row_end = min(p.M, row0 + rows_per_block);

  // ---- stage the block's X rows (columns >= K zeroed: the padding may hold
  // anything, and 0 * NaN would poison the sums)
  for (int i = tid; i < rows_per_block * KV; i += SK_THREADS) {
    const int r = i / KV, v = i % KV;
    const int m = min(row0 + r, p.M - 1);
    const int64_t src = p.idx ? (int64_t)p.idx[m] : (int64_t)m;
    float4 x = *reinterpret_cast<const float4*>(p.X + src * p.ldx + 4 * v);
    x.x = (4 * v + 0 < p.K) ? x.x : 0.f;
    x.y = (4 * v + 1 < p.K) ? x.y : 0.f;
    x.z = (4 * v + 2 < p.K) ? x.z : 0.f;
    x.w = (4 * v + 3 < p.K) ? x.w : 0.f;
    *reinterpret_cast<float4*>(xs + (r * KV + v) * 4) = x;
  }

  // wk[k] = W(n0..n0+3, k)
  float4 wk[4 * KV];
  if (W_KC) {
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      float4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[j] = *reinterpret_cast<const float4*>(p.W + (int64_t)(nc + j) * p.ldw + 4 * v);
      wk[4 * v + 0] = make_float4(r[0].x, r[1].x, r[2].x, r[3].x);
      wk[4 * v + 1] = make_float4(r[0].y, r[1].y, r[2].y, r[3].y);
      wk[4 * v + 2] = make_float4(r[0].z, r[1].z, r[2].z, r[3].z);
      wk[4 * v + 3] = make_float4(r[0].w, r[1].w, r[2].w, r[3].w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4 * KV; ++k)
      wk[k] = *reinterpret_cast<const float4*>(p.W + (int64_t)min(k, p.K - 1) * p.ldw + nc);
  }
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == 0 && p.bias) bias = *reinterpret_cast<const float4*>(p.bias + nc);
  __syncthreads();

  for (int it = 0; it < p.rows_per_thread; it += SK_UNROLL) {
    float4 h[SK_UNROLL];
    if (EPI == 1) {
#pragma unroll
      for (int u = 0; u < SK_UNROLL; ++u) {
        const int m = min(row0 + (it + u) * n_rg + rg, p.M - 1);
        h[u] = *reinterpret_cast<const float4*>(p.H + (int64_t)m * p.ldh + nc);
      }
    }
#pragma unroll
    for (int u = 0; u < SK_UNROLL; ++u) {
      const int rl = (it + u) * n_rg + rg;
      const int m = row0 + rl;
      // two packed FMAs (v_pk_fma_f32) per k instead of four scalar ones: the same
      // fmaf chain per output, half the vector-issue slots
      const float4 a0 = (EPI == 0) ? bias : make_float4(0.f, 0.f, 0.f, 0.f);
      sk_v2f alo = {a0.x, a0.y}, ahi = {a0.z, a0.w};
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + (rl * KV + v) * 4);
        const float xk[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 w = wk[4 * v + j];
          const sk_v2f xx = {xk[j], xk[j]};
          const sk_v2f wlo = {w.x, w.y}, whi = {w.z, w.w};
          alo = __builtin_elementwise_fma(xx, wlo, alo);
          ahi = __builtin_elementwise_fma(xx, whi, ahi);
        }
      }
      float4 a = make_float4(alo.x, alo.y, ahi.x, ahi.y);
      if (EPI == 0) {
        if (p.act == 1) {
          a.x = tanh_fast(a.x); a.y = tanh_fast(a.y);
          a.z = tanh_fast(a.z); a.w = tanh_fast(a.w);
        }
      } else {
        a.x *= (1.f - h[u].x * h[u].x);
        a.y *= (1.f - h[u].y * h[u].y);
        a.z *= (1.f - h[u].z * h[u].z);
        a.w *= (1.f - h[u].w * h[u].w);
      }
      if (m < row_end && col_ok)
        *reinterpret_cast<float4*>(p.Y + (int64_t)m * p.ldy + n0) = a;
    }
  }
}

// ---------------------------------------------------------------------------
// C(wide c, narrow j) = sum_r Wd[row_w(r), c] * Nr[row_n(r), j] per split of rows
// ---------------------------------------------------------------------------
struct SkinnyWgradParams {
  const float* Wd;       // wide operand  [rows][ldw]
  int64_t ldw;
  const int32_t* w_idx;
  const float* Nr;       // narrow operand [rows][ldn], NS valid floats
  int64_t ldn;
  const int32_t* n_idx;
  int rows, wide, NS;
  int rows_per_split;
  int qpr;               // wide column quads per workgroup (<= 64)
  int n_colblk, n_splits;  // logical grid; launched 1-D in XCD-aware order
  float* C;
  int64_t c_wide_stride, c_narrow_stride;  // C(c, j) at c * cws + j * cns
  int64_t split_stride;
  float* colsum_wide;    // optional: sum_r Wd[., c]  -> [split][c]
  float* colsum_narrow;  // optional: sum_r Nr[., j]  -> [split][j]
  // DZ instantiation (head weight gradient + the data gradient of the layer below
  // in one pass over H): dz_out[r, c] = (sum_j Nr[r, j] * Wn[j, c]) * (1 - Wd[r, c]^2)
  const float* Wn;       // [NS][ldwn] head weights, wide-contiguous
  int64_t ldwn;
  float* dz_out;         // [rows][lddz]
  int64_t lddz;
};

constexpr int SW_THREADS = 256;
constexpr int SW_QPR = 16;             // column quads (64 floats) per workgroup
constexpr int SW_CHUNK = 256;          // rows of the narrow operand staged at a time
constexpr int SW_LDS_FLOATS = 8192;    // 32 KB: narrow stage, then reduction stage

// One workgroup = (64 wide columns) x (one split of the rows): 16 row groups of
// 16 threads; a thread owns 4 columns and keeps 2 x UNROLL 16-B loads of the wide
// operand in flight (the loads of the next step are issued before the FMAs of
// the current one).  grid = (wide / 64, n_splits): 512 workgroups at the C3
// minibatch, several per CU.
// NV = round4(NS) / 4; NSUM: also the column sums of the narrow operand; DZ: also
// the data gradient of the layer below (the wide operand is then its tanh output)
template <int NV, bool NSUM, bool DZ = false>
// (two waves per SIMD: a streaming kernel lives on the loads it has in flight; left
// alone the compiler takes 280-350 registers for the DZ variants: one wave per SIMD)
__global__ __launch_bounds__(SW_THREADS, (DZ && NV >= 4) ? 1 : 2) void skinny_wgrad_kernel(
    SkinnyWgradParams p) {
  constexpr int U = NV <= 2 ? (DZ ? 4 : 8) : 4;
  __shared__ __attribute__((aligned(16))) float red[SW_LDS_FLOATS];
  const int tid = threadIdx.x;
  const int q = tid % p.qpr;
  const int rg = tid / p.qpr;
  const int n_rg = SW_THREADS / p.qpr;
  // the column blocks of a split read the same narrow rows: same XCD (common.h)
  int split, colblk;
  ga_xcd_group((int)blockIdx.x, p.n_splits, p.n_colblk, &split, &colblk);
  const int c0 = (colblk * p.qpr + q) * 4;
  const bool col_ok = c0 < p.wide;
  const int cc = col_ok ? c0 : 0;
  const int r_beg = split * p.rows_per_split;
  const int r_end = min(p.rows, r_beg + p.rows_per_split);

  float4 acc[4 * NV];   // acc[j] = C(c0..c0+3, j)
  float4 wsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float nsum[NSUM ? 4 * NV : 1];
#pragma unroll
  for (int j = 0; j < 4 * NV; ++j) {
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (NSUM) nsum[j] = 0.f;
  }
  const bool want_nsum = NSUM && colblk == 0 && q == 0;
  float4 wq[DZ ? 4 * NV : 1];  // head weights of this thread's 4 columns
  if (DZ) {
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j)
      wq[j] = (j < p.NS) ? *reinterpret_cast<const float4*>(p.Wn + (int64_t)j * p.ldwn + cc)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  for (int rc = r_beg; rc < r_end; rc += SW_CHUNK) {
    const int rc_end = min(r_end, rc + SW_CHUNK);
    float4 w[U], wn[U];
    // first step's wide loads go out before the narrow stage is waited for
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(rc + rg + u * n_rg, rc_end - 1);
      const int64_t rw = p.w_idx ? (int64_t)p.w_idx[r] : (int64_t)r;
      w[u] = *reinterpret_cast<const float4*>(p.Wd + rw * p.ldw + cc);
    }
    __syncthreads();
    for (int i = tid; i < SW_CHUNK * NV; i += SW_THREADS) {
      const int r = i / NV, v = i % NV;
      const int row = min(rc + r, r_end - 1);
      const int64_t src = p.n_idx ? (int64_t)p.n_idx[row] : (int64_t)row;
      float4 nv4 = *reinterpret_cast<const float4*>(p.Nr + src * p.ldn + 4 * v);
      if (DZ) {  // padding columns may hold anything: they must not reach dz
        nv4.x = (4 * v + 0 < p.NS) ? nv4.x : 0.f;
        nv4.y = (4 * v + 1 < p.NS) ? nv4.y : 0.f;
        nv4.z = (4 * v + 2 < p.NS) ? nv4.z : 0.f;
        nv4.w = (4 * v + 3 < p.NS) ? nv4.w : 0.f;
      }
      *reinterpret_cast<float4*>(red + (r * NV + v) * 4) = nv4;
    }
    __syncthreads();
    for (int rb = rc + rg; rb < rc_end; rb += n_rg * U) {
      const int rb_next = rb + n_rg * U;
      if (rb_next < rc_end) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = min(rb_next + u * n_rg, rc_end - 1);
          const int64_t rw = p.w_idx ? (int64_t)p.w_idx[r] : (int64_t)r;
          wn[u] = *reinterpret_cast<const float4*>(p.Wd + rw * p.ldw + cc);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rb + u * n_rg;
        const bool live = r < rc_end;
        const int rl = min(r, rc_end - 1) - rc;
        const float4 wv = live ? w[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        wsum.x += wv.x; wsum.y += wv.y; wsum.z += wv.z; wsum.w += wv.w;
        float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float4 nv = *reinterpret_cast<const float4*>(red + (rl * NV + v) * 4);
          const float ns[4] = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4& a = acc[4 * v + j];
            a.x = fmaf(wv.x, ns[j], a.x);
            a.y = fmaf(wv.y, ns[j], a.y);
            a.z = fmaf(wv.z, ns[j], a.z);
            a.w = fmaf(wv.w, ns[j], a.w);
            if (NSUM && want_nsum && live) nsum[4 * v + j] += ns[j];
            if (DZ) {
              const float4 wn = wq[4 * v + j];
              dz.x = fmaf(ns[j], wn.x, dz.x);
              dz.y = fmaf(ns[j], wn.y, dz.y);
              dz.z = fmaf(ns[j], wn.z, dz.z);
              dz.w = fmaf(ns[j], wn.w, dz.w);
            }
          }
        }
        if (DZ && live && col_ok) {
          dz.x *= (1.f - wv.x * wv.x); dz.y *= (1.f - wv.y * wv.y);
          dz.z *= (1.f - wv.z * wv.z); dz.w *= (1.f - wv.w * wv.w);
          *reinterpret_cast<float4*>(p.dz_out + (int64_t)r * p.lddz + c0) = dz;
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads row by row
      }
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = wn[u];
    }
  }

  // ---- sum the row groups in a fixed order through LDS, a few j at a time
  float* Cs = p.C + (int64_t)split * p.split_stride;
  const int quads = p.qpr;                         // float4 per (group, j)
  const int per_j = n_rg * quads * 4;              // = 1024 floats staged per j
  const int jp = SW_LDS_FLOATS / per_j;            // j per pass
  constexpr int n_j = 4 * NV + 1;                  // + 1: the wide column sums
  for (int j0 = 0; j0 < n_j; j0 += jp) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < n_j; ++j) {
      if (j >= j0 && j < j0 + jp) {
        const float4 v = (j < 4 * NV) ? acc[j < 4 * NV ? j : 0] : wsum;
        *reinterpret_cast<float4*>(red + ((j - j0) * n_rg + rg) * quads * 4 + q * 4) = v;
      }
    }
    __syncthreads();
    const int jn = min(jp, n_j - j0);
    for (int o = tid; o < jn * quads; o += SW_THREADS) {
      const int jj = o / quads, qq = o % quads;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int g = 0; g < n_rg; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(
            red + (jj * n_rg + g) * quads * 4 + qq * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int j = j0 + jj;
      const int c = (colblk * p.qpr + qq) * 4;
      if (c >= p.wide) continue;
      if (j < 4 * NV) {
        if (j >= p.NS) continue;
        if (p.c_wide_stride == 1) {
          *reinterpret_cast<float4*>(Cs + (int64_t)j * p.c_narrow_stride + c) = s;
        } else {
          Cs[(int64_t)(c + 0) * p.c_wide_stride + j * p.c_narrow_stride] = s.x;
          Cs[(int64_t)(c + 1) * p.c_wide_stride + j * p.c_narrow_stride] = s.y;
          Cs[(int64_t)(c + 2) * p.c_wide_stride + j * p.c_narrow_stride] = s.z;
          Cs[(int64_t)(c + 3) * p.c_wide_stride + j * p.c_narrow_stride] = s.w;
        }
      } else if (p.colsum_wide) {
        *reinterpret_cast<float4*>(p.colsum_wide + (int64_t)split * p.split_stride + c) = s;
      }
    }
  }
  // ---- narrow column sums: thread q == 0 of every row group holds a partial
  if (NSUM && colblk == 0) {
    __syncthreads();
    if (q == 0) {
#pragma unroll
      for (int j = 0; j < 4 * NV; ++j) red[rg * 4 * NV + j] = nsum[j];
    }
    __syncthreads();
    if (tid < p.NS) {
      float s = 0.f;
      for (int g = 0; g < n_rg; ++g) s += red[g * 4 * NV + tid];
      p.colsum_narrow[(int64_t)split * p.split_stride + tid] = s;
    }
  }
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// columns per workgroup: the whole row when it is a power of two <= 256 floats,
// else 256-float column blocks
inline int pick_qpr(int wide) {
  if (wide % 4 != 0) return 0;
  if (wide <= 256) return pow2(wide / 4) ? wide / 4 : 0;
  return (wide % 256 == 0) ? 64 : 0;
}

template <int KV>
void launch_fwd(const SkinnyFwdParams& p, bool w_kc, int epi, dim3 grid,
                hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
  const int n_rg = SK_THREADS / p.qpr;
  const unsigned lds = (unsigned)(n_rg * p.rows_per_thread * KV * 16);
  if (w_kc && epi == 0)
    hipExtLaunchKernelGGL((skinny_fwd_kernel<KV, true, 0>), grid, dim3(SK_THREADS), lds,
                          stream, e0, e1, 0, p);
  else if (!w_kc && epi == 1)
    hipExtLaunchKernelGGL((skinny_fwd_kernel<KV, false, 1>), grid, dim3(SK_THREADS), lds,
                          stream, e0, e1, 0, p);
}

}  // namespace

// Internal entry points (called from gemm.hip's layer dispatch; not in the C ABI).
// Return 1 when the shape is not one these kernels take (caller falls back to the
// MFMA tile kernel), 0 on launch, negative on error.

// Y = act(X W^T + b): W[n][ldw] k-contiguous.  epi 0.
// Y = (X W) * (1 - H^2): W[k][ldw] n-contiguous.  epi 1.
int ga_skinny_forward(const float* X, int64_t ldx, const int32_t* idx, const float* W,
                      int64_t ldw, bool w_kc, const float* bias, int act,
                      const float* H, int64_t ldh, float* Y, int64_t ldy, int M, int N,
                      int K, hipStream_t stream) {
  const int qpr = pick_qpr(N);
  if (K < 1 || K > 32 || qpr == 0 || ldx % 4 != 0 || ldw % 4 != 0 || ldy % 4 != 0 ||
      (H && ldh % 4 != 0) || !ga_aligned16(X) || !ga_aligned16(W) || !ga_aligned16(Y) ||
      (bias && !ga_aligned16(bias)) || (H && !ga_aligned16(H)) || ldx < ((K + 3) & ~3))
    return 1;
  if (w_kc != (H == nullptr)) return 1;  // only the two layer products above
  SkinnyFwdParams p;
  p.X = X; p.ldx = ldx; p.idx = idx; p.W = W; p.ldw = ldw; p.bias = bias; p.H = H;
  p.ldh = ldh; p.Y = Y; p.ldy = ldy; p.M = M; p.N = N; p.K = K; p.act = act;
  p.qpr = qpr;
  const int n_rg = SK_THREADS / qpr;
  // 32 rows per workgroup (never fewer than one unroll per thread): two generations
  // of workgroups per CU, so one's prologue (gathered X rows, 20 KB of W quads from
  // L2) runs under the other's stores.  Round 1 preferred 128 rows while the other
  // update chain's streaming kernels shared the chip (146.2 / 143.3 / 152.9 ms per
  // C3 iteration at 32 / 128 / 256); with those kernels folded into the GEMMs it
  // is 131.5 / 133.0 / 132.7 ms at 32 / 64 / 128 (18.5 / 18.9 / 20.2 us a launch).
  // Packed FMAs changed nothing: the kernel is bound by its store stream.
  // (... but at least one workgroup per CU)
  static int rows_wg_env = -1;  // A/B runs: GARAGE_AMD_SKINNY_ROWS
  if (rows_wg_env < 0) {
    const char* e = getenv("GARAGE_AMD_SKINNY_ROWS");
    rows_wg_env = e ? atoi(e) : 0;
  }
  int rows_wg = rows_wg_env > 0 ? rows_wg_env : 32;
  while (rows_wg > 32 && (int64_t)M < 256 * (int64_t)rows_wg) rows_wg >>= 1;
  p.rows_per_thread = ((rows_wg / n_rg + SK_UNROLL - 1) / SK_UNROLL) * SK_UNROLL;
  if (p.rows_per_thread < SK_UNROLL) p.rows_per_thread = SK_UNROLL;
  const int rows_per_block = n_rg * p.rows_per_thread;
  if ((int64_t)rows_per_block * ((K + 3) / 4) * 16 > 48 * 1024) return 1;
  dim3 grid((unsigned)ga_ceil_div(M, rows_per_block),
            (unsigned)ga_ceil_div(N, 4 * qpr));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // algorithmic bytes: the [M x N] output (+ H for the data gradient) + X
  const double bytes = 4.0 * M * ((double)N * (H ? 2 : 1) + K);
  ga_prof_events(GA_PROF_SKINNY_FWD, bytes, &e0, &e1);
  const int epi = H ? 1 : 0;
  switch ((K + 3) / 4) {
    case 1: launch_fwd<1>(p, w_kc, epi, grid, stream, e0, e1); break;
    case 2: launch_fwd<2>(p, w_kc, epi, grid, stream, e0, e1); break;
    case 3: launch_fwd<3>(p, w_kc, epi, grid, stream, e0, e1); break;
    case 4: launch_fwd<4>(p, w_kc, epi, grid, stream, e0, e1); break;
    case 5: launch_fwd<5>(p, w_kc, epi, grid, stream, e0, e1); break;
    case 6: launch_fwd<6>(p, w_kc, epi, grid, stream, e0, e1); break;
    case 7: launch_fwd<7>(p, w_kc, epi, grid, stream, e0, e1); break;
    default: launch_fwd<8>(p, w_kc, epi, grid, stream, e0, e1); break;
  }
  GA_CHECK_LAUNCH("skinny_fwd");
  return GA_OK;
}

int ga_skinny_wgrad(const float* Wd, int64_t ldw, const int32_t* w_idx, const float* Nr,
                    int64_t ldn, const int32_t* n_idx, int rows, int wide, int NS,
                    int rows_per_split, int n_splits, float* C, int64_t c_wide_stride,
                    int64_t c_narrow_stride, int64_t split_stride, float* colsum_wide,
                    float* colsum_narrow, const float* Wn, int64_t ldwn, float* dz_out,
                    int64_t lddz, hipStream_t stream) {
  // whole rows of <= 64 floats (a power of two), else 64-float column blocks
  const int qpr = (wide % 4 != 0) ? 0
                  : (wide <= 4 * SW_QPR ? (pow2(wide / 4) ? wide / 4 : 0) : SW_QPR);
  // (16 NV accumulator registers per thread: beyond NS = 24 the kernel spills)
  if (NS < 1 || NS > 24 || qpr == 0 || ldw % 4 != 0 || ldn % 4 != 0 ||
      ldn < ((NS + 3) & ~3) || !ga_aligned16(Wd) || !ga_aligned16(Nr) ||
      split_stride % 4 != 0 || (c_wide_stride == 1 && (c_narrow_stride % 4 != 0 ||
                                                       !ga_aligned16(C))) ||
      (colsum_wide && !ga_aligned16(colsum_wide)))
    return 1;
  SkinnyWgradParams p;
  p.Wd = Wd; p.ldw = ldw; p.w_idx = w_idx; p.Nr = Nr; p.ldn = ldn; p.n_idx = n_idx;
  p.rows = rows; p.wide = wide; p.NS = NS; p.rows_per_split = rows_per_split;
  p.qpr = qpr; p.C = C; p.c_wide_stride = c_wide_stride;
  p.c_narrow_stride = c_narrow_stride; p.split_stride = split_stride;
  p.colsum_wide = colsum_wide; p.colsum_narrow = colsum_narrow;
  p.Wn = Wn; p.ldwn = ldwn; p.dz_out = dz_out; p.lddz = lddz;
  const bool dzf = dz_out != nullptr;
  if (dzf && (w_idx || !Wn || ldwn % 4 != 0 || lddz % 4 != 0 || !ga_aligned16(Wn) ||
              !ga_aligned16(dz_out) || NS > 16 || colsum_narrow == nullptr))
    return 1;
  p.n_colblk = (int)ga_ceil_div(wide, 4 * qpr);
  p.n_splits = n_splits;
  dim3 grid((unsigned)(p.n_colblk * p.n_splits));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // algorithmic bytes: both operands once (+ the data gradient written once)
  const double bytes = 4.0 * rows * ((double)wide * (dzf ? 2 : 1) + NS);
  ga_prof_events(GA_PROF_SKINNY_WGRAD, bytes, &e0, &e1);
  const bool nsum = colsum_narrow != nullptr;
#define GA_SW_CASE(NVV)                                                              \
  case NVV:                                                                          \
    if (dzf && NVV <= 4)                                                             \
      hipExtLaunchKernelGGL((skinny_wgrad_kernel<(NVV <= 4 ? NVV : 1), true, true>), \
                            grid, dim3(SW_THREADS), 0, stream, e0, e1, 0, p);        \
    else if (nsum)                                                                   \
      hipExtLaunchKernelGGL((skinny_wgrad_kernel<NVV, true>), grid, dim3(SW_THREADS), \
                            0, stream, e0, e1, 0, p);                                \
    else                                                                             \
      hipExtLaunchKernelGGL((skinny_wgrad_kernel<NVV, false>), grid,                 \
                            dim3(SW_THREADS), 0, stream, e0, e1, 0, p);              \
    break;
  switch ((NS + 3) / 4) {
    GA_SW_CASE(1) GA_SW_CASE(2) GA_SW_CASE(3) GA_SW_CASE(4) GA_SW_CASE(5) GA_SW_CASE(6)
    default: return 1;
  }
#undef GA_SW_CASE
  GA_CHECK_LAUNCH("skinny_wgrad");
  return GA_OK;
}
