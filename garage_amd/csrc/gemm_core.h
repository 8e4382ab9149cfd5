// The k-loop shared by the fp32 MFMA GEMM kernels (gemm.hip) and the fused
// training-step kernels built on it (fused_train.hip): operand-tile loader
// (global -> registers -> LDS) and the MFMA main loop.  See gemm.hip for the
// layout notes.
#pragma once
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int PAD = 4;
// One LDS stage per operand tile (37 KB per 128x128 workgroup): a second stage
// (one barrier per k-step instead of two) measured no faster at the K = 256 shapes
// of this workload, and the small footprint lets the workgroups of the policy and
// the value-function update chains co-reside on a CU when they run on two streams.

enum Epilogue { EPI_BIAS_ACT = 0, EPI_MUL_DTANH = 1, EPI_PLAIN = 2 };

// tanh on the hardware exp / rcp units (common.h: the one definition every kernel
// shares; 64 of these per lane per output tile).
__device__ __forceinline__ float tanh_fast(float x) {
  return ga_tanh(x);  // common.h
}

// Activations (the reference accepts any callable as hidden / output
// nonlinearity, torch/modules/multi_headed_mlp_module.py:154-197; these are the ones
// whose slope is a function of the OUTPUT, so that the backward pass needs no
// pre-activation in memory).
// Forward code (GemmParams::act, ga_mlp_desc::output_act): 0 none, 1 tanh, 2 relu,
//   3 sigmoid, 4 elu (alpha 1), 5 leaky_relu (slope 0.01), 6 softplus (beta 1,
//   threshold 20) -- torch.nn.functional's defaults.
// Network code (ga_mlp_desc::hidden_act, GemmParams::hact): 0 tanh (what a zeroed
//   descriptor means), 1 relu, 2 none, 3 .. 6 as above.
// (the codes beyond tanh / relu / none sit behind a wave-uniform branch and a call:
// inlined into a select chain their exp / log expansions ran for EVERY element of
// every epilogue -- C5's forward GEMM 230 -> 235 us)
__device__ __attribute__((noinline)) float act_apply_more(float v, int act) {
  switch (act) {
    case 3: return 1.f / (1.f + expf(-v));
    case 4: return v > 0.f ? v : expm1f(v);
    case 5: return v > 0.f ? v : 0.01f * v;
    default: return v > 20.f ? v : log1pf(expf(v));  // 6 softplus
  }
}
__device__ __forceinline__ float act_apply(float v, int act) {
  if (__builtin_expect(act <= 2, 1))
    return act == 1 ? tanh_fast(v) : (act == 2 ? fmaxf(v, 0.f) : v);
  return act_apply_more(v, act);
}
__device__ __attribute__((noinline)) float act_slope_more(float h, int hact) {
  switch (hact) {
    case 3: return h * (1.f - h);
    case 4: return h > 0.f ? 1.f : h + 1.f;       // exp(x) = h + 1 for x <= 0
    case 5: return h > 0.f ? 1.f : 0.01f;
    default: return 1.f - expf(-h);                // 6: sigmoid(x), h = log(1 + e^x)
  }
}
__device__ __forceinline__ float act_slope(float h, int hact) {
  if (__builtin_expect(hact <= 2, 1))
    return hact == 0 ? 1.f - h * h : (hact == 1 ? (h > 0.f ? 1.f : 0.f) : 1.f);
  return act_slope_more(h, hact);
}
__host__ __device__ inline int act_forward_code(int hidden_act) {
  return hidden_act == 0 ? 1 : (hidden_act == 1 ? 2 : (hidden_act == 2 ? 0 : hidden_act));
}
// slope of an OUTPUT activation given in forward code
__device__ __forceinline__ float act_slope_fwd(float o, int act) {
  return act_slope(o, act == 0 ? 2 : (act == 1 ? 0 : (act == 2 ? 1 : act)));
}

// ---- split-operand k-loop (opt-in: ga_set_split_bf16 / GARAGE_AMD_SPLIT_BF16=1).
// v_mfma_f32_32x32x2_f32 runs on the vector ALU's fp32 lanes (tools/mfma_valu_overlap:
// every vector instruction beside it costs its full issue time, 64 cycles per
// 32x32x2 on top); v_mfma_f32_32x32x16_bf16 is 16x the rate on the matrix unit proper
// and vector work DOES issue beside it.  An fp32 value is split EXACTLY into three
// bf16 terms x = hi + mid + lo (8 + 8 + 8 significant bits, each the truncation of
// the remainder); a product a b is the six bf16 products a_i b_j with i + j <= 2
// (the three dropped ones are below 2^-24 |a b|), accumulated in fp32 by the MFMA,
// smallest terms first: 6 x 32 cycles for 16 k against 8 x 64 for exact fp32.
typedef __bf16 ft_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ft_u32x4 __attribute__((ext_vector_type(4)));
constexpr int FT_PLANE_ROW_B = 80;  // bytes per LDS plane row: 32 bf16 + 16 (conflict free)
constexpr int FT_PLANE_B = 64 * FT_PLANE_ROW_B;  // 64-row tiles
constexpr int FT_ABUF_B = 3 * FT_PLANE_B;  // hi, mid, lo of one 64 x 32 operand chunk

typedef __bf16 ft_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ft_f32x2 __attribute__((ext_vector_type(2)));
// (bf16(b) : bf16(a)), round to nearest even -- one v_cvt_pk_bf16_f32; a is the lower k
__device__ __forceinline__ uint32_t ft_rne_pack(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(ft_f32x2{a, b}, ft_bf16x2));
}
// Two values (consecutive k) -> the packed hi / mid / lo terms.  Each term is the
// NEAREST bf16 of what the terms before it left, so the remainders -- and with them the
// three dropped products -- carry either sign: truncation instead makes every term share
// the sign of x and every dropped product the sign of a b, a relative bias of ~2^-23
// that sums coherently (measured: the bias-gradient column sums were 4x further from
// fp64 than the exact fp32 kernel's; with rounding they are as close).  The two
// remainders are exact in fp32; the last rounding is below 2^-26 |x|.
__device__ __forceinline__ void ft_split3_pair(float x0, float x1, uint32_t& hi,
                                               uint32_t& mid, uint32_t& lo) {
  hi = ft_rne_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16);
  const float r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = ft_rne_pack(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16);
  const float s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = ft_rne_pack(s0, s1);
}
// the six products of one 32 x 32 x 16 block, small to large
__device__ __forceinline__ void ft_mfma6(const ft_bf16x8 (&a)[3], const ft_bf16x8 (&b)[3],
                                         f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}


struct GemmParams {
  const float* A;
  int64_t lda;           // floats between consecutive memory lines of A
  const int32_t* a_idx;  // optional gather applied to A's memory-line index
  const float* B;
  int64_t ldb;
  const int32_t* b_idx;
  float* C;
  int64_t c_rs, c_cs;    // C(m,n) at C[m * c_rs + n * c_cs]
  int M, N, K;
  int epi;
  const float* bias;     // EPI_BIAS_ACT: per-n bias (may be null)
  int act;               // 0 identity, 1 tanh
  const float* H;        // EPI_MUL_DTANH (or EPI_BIAS_ACT with H set): activation
  int64_t ldh;           // outputs H[m * ldh + n]; the result is scaled by the
  int hact;              // activation's slope there (network code, 0 = tanh)
  int accum;             // 1: add the product to what C already holds
  int k_per_split;       // multiple of BK
  int64_t c_split_stride;
  float* colsum;         // optional: sum_k of operand A (or B) -> colsum[line]
  int colsum_of_b;       // 0: columns of A tile (index m), 1: of B tile (index n)
  int64_t colsum_split_stride;
  int gx, gy, gz;        // logical grid (m blocks, n blocks, splits); 1-D launch
  // HEAD kernels (the tile spans all N columns): the next, narrow layer is applied to
  // the staged output rows in the epilogue: head_out[m, j] = head_bias[j] +
  // sum_n C(m, n) * head_W[j * head_ldw + n],  j < head_n <= 8
  const float* head_W;
  int64_t head_ldw;
  const float* head_bias;
  int head_n;
  float* head_out;
  int64_t head_ld;
  // split-operand instantiation (opt-in): the B operand as three bf16 planes in
  // fragment order (fused_train.h: ga_weight_planes), plane pl at + pl * stride,
  // bplane_nblk = round32(N) / 32 column blocks per 16-deep k group
  const uint16_t* bplanes;
  int64_t bplane_stride;
  int bplane_nblk;
};

// One [BR x BK] operand tile: global -> registers -> LDS.
//   KC = true : memory line = r (tile row), contiguous along k
//   KC = false: memory line = k,            contiguous along r
// Every global load is UNCONDITIONAL (indices are clamped into valid memory and
// the out-of-range lanes are zeroed when the registers are written to LDS):
// a load inside a data-dependent branch makes hipcc wait vmcnt(0) right behind
// it, which serialises the whole tile fetch in front of the MFMAs.  The gathered
// line numbers (`idx`) are fetched one tile ahead for the same reason.
// FULL: the workgroup's tile rows are entirely inside the matrices, so the clamps
// and the zero masks (16 v_cndmask per vector pair and k-step) drop out on every
// k-step but the last (which may be partial).
template <int BR, bool KC, int NT, int BKT, bool FULL>
struct TileLoader {
  static constexpr int NV = BR * BKT / 4 / NT;  // float4 per thread
  static constexpr int VPR = BKT / 4;           // vectors per row (KC = true)
  static constexpr int VPL = BR / 4;            // vectors per line (KC = false)
  float4 regs[NV];
  int32_t cur[NV];  // memory line (after the optional gather) of each vector
  int32_t nxt[NV];
  // KC = false without a gather: a vector's address advances by BKT lines per step --
  // a pointer per vector set up once (init_linear) and ONE 64-bit add per load replace
  // the per-step line clamps and the 64-bit line * ld products (two quarter-rate 32-bit
  // multiplies and a mad per vector: on this chip every vector instruction beside an
  // fp32 MFMA is matrix time lost)
  const float* lin_ptr[NV];
  bool lin;

  // lines of the first tile (KC: the rows, fixed for the whole kernel)
  __device__ __forceinline__ void init(const int32_t* __restrict__ idx, int r0,
                                       int R, int kbeg, int kend) {
    const int tid = threadIdx.x;
    int want[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + NT * i;
      want[i] = KC ? min(r0 + f / VPR, R - 1) : min(kbeg + f / VPL, kend - 1);
      want[i] = max(want[i], 0);
    }
    if (idx) {
#pragma unroll
      for (int i = 0; i < NV; ++i) cur[i] = idx[want[i]];
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) cur[i] = want[i];
    }
  }

  // (FULL, KC = false, no gather) pointers of the first tile; `base` + r0 as in load()
  __device__ __forceinline__ void init_linear(const float* __restrict__ base, int64_t ld,
                                              const int32_t* __restrict__ idx, int r0,
                                              int kbeg) {
    lin = FULL && !KC && idx == nullptr;
    if (!lin) return;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + NT * i;
      lin_ptr[i] = base + (int64_t)(kbeg + f / VPL) * ld + r0 + 4 * (f % VPL);
    }
  }

  // KC = false only: lines of the tile that starts at k0 (one tile ahead)
  __device__ __forceinline__ void prefetch_lines(const int32_t* __restrict__ idx,
                                                 int k0, int kend) {
    if (KC) return;
    if (FULL && lin && k0 + BKT < kend) return;  // the linear path needs no line numbers
    const int tid = threadIdx.x;
    if (idx) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        nxt[i] = idx[max(min(k0 + (tid + NT * i) / VPL, kend - 1), 0)];
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        nxt[i] = max(min(k0 + (tid + NT * i) / VPL, kend - 1), 0);
    }
  }

  __device__ __forceinline__ void rotate() {
    if (KC) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
  }

  // span = number of valid floats along the contiguous direction (K or R)
  // tail: this is the (possibly partial) last k-step of the block's k range; a
  // FULL loader still clamps / masks there, so only the tile's row range has to
  // be interior for the fast path, not its k range
  __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld,
                                       int r0, int k0, int span, bool tail, int kbeg = 0) {
    const int tid = threadIdx.x;
    if (FULL && !KC && lin && !tail) {
      const int64_t off = (int64_t)(k0 - kbeg) * ld;  // wave-uniform
#pragma unroll
      for (int i = 0; i < NV; ++i)
        regs[i] = *reinterpret_cast<const float4*>(lin_ptr[i] + off);
      return;
    }
    const int last = max(((span + 3) & ~3) - 4, 0);  // last in-bounds vector
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + NT * i;
      const int c = KC ? (k0 + 4 * (f % VPR)) : (r0 + 4 * (f % VPL));
      regs[i] = *reinterpret_cast<const float4*>(base + (int64_t)cur[i] * ld +
                                                 ((FULL && !tail) ? c : min(c, last)));
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ tile, int r0, int R,
                                        int k0, int kend, bool tail) const {
    constexpr int LD = BR + PAD;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + NT * i;
      float4 v = regs[i];
      if (KC) {
        const int r = f / VPR;
        const int k = 4 * (f % VPR);
        if (!FULL || tail) {
          const bool row_ok = (r0 + r) < R;
          v.x = (row_ok && k0 + k + 0 < kend) ? v.x : 0.f;
          v.y = (row_ok && k0 + k + 1 < kend) ? v.y : 0.f;
          v.z = (row_ok && k0 + k + 2 < kend) ? v.z : 0.f;
          v.w = (row_ok && k0 + k + 3 < kend) ? v.w : 0.f;
        }
        // k-contiguous in memory stays k-contiguous in LDS: [BR][BK + PAD],
        // one ds_write_b128 per vector, conflict free (8 lanes = one 128-B row)
        *reinterpret_cast<float4*>(tile + r * (BKT + PAD) + k) = v;
      } else {
        const int k = f / VPL;
        const int r = 4 * (f % VPL);
        if (!FULL || tail) {
          const bool k_ok = (k0 + k) < kend;
          v.x = (k_ok && r0 + r + 0 < R) ? v.x : 0.f;
          v.y = (k_ok && r0 + r + 1 < R) ? v.y : 0.f;
          v.z = (k_ok && r0 + r + 2 < R) ? v.z : 0.f;
          v.w = (k_ok && r0 + r + 3 < R) ? v.w : 0.f;
        }
        *reinterpret_cast<float4*>(tile + k * LD + r) = v;
      }
    }
  }
};

// The k-loop of one workgroup: global -> registers -> LDS -> MFMA.  FULL selects the
// mask-free loader (the caller has checked that this workgroup's tile rows are
// entirely inside the matrices; the last k-step is masked either way).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC, int BKT,
          bool FULL, int TM, int TN>
__device__ __forceinline__ void gemm_mainloop(const GemmParams& p, float* lds,
                                              f32x16 (&acc)[TM][TN], float& csum,
                                              bool do_colsum, int m0, int n0,
                                              int kbeg, int kend, int wm0, int wn0) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD, LDK = BKT + PAD;
  constexpr int A_FLOATS = A_KC ? BM * LDK : BKT * LDA_S;
  const int lane = threadIdx.x & 63;
  TileLoader<BM, A_KC, NT, BKT, FULL> la;
  TileLoader<BN, B_KC, NT, BKT, FULL> lb;
  const int a_span = A_KC ? p.K : p.M;  // valid floats along the contiguous axis
  const int b_span = B_KC ? p.K : p.N;
  const int nk = (kend - kbeg + BKT - 1) / BKT;
  float* As = lds;
  float* Bs = lds + A_FLOATS;
  if (nk > 0) {
    la.init(p.a_idx, m0, p.M, kbeg, kend);
    lb.init(p.b_idx, n0, p.N, kbeg, kend);
    la.init_linear(p.A, p.lda, p.a_idx, m0, kbeg);
    lb.init_linear(p.B, p.ldb, p.b_idx, n0, kbeg);
    const bool t0 = nk == 1;
    la.load(p.A, p.lda, m0, kbeg, a_span, t0, kbeg);
    lb.load(p.B, p.ldb, n0, kbeg, b_span, t0, kbeg);
    la.prefetch_lines(p.a_idx, kbeg + BKT, kend);
    lb.prefetch_lines(p.b_idx, kbeg + BKT, kend);
    la.store(As, m0, p.M, kbeg, kend, t0);
    lb.store(Bs, n0, p.N, kbeg, kend, t0);
  }
  __syncthreads();

  const int half = lane >> 5, l31 = lane & 31;
  for (int s = 0; s < nk; ++s) {
    const bool more = (s + 1 < nk);
    const bool tail = (s + 2 == nk);  // the tile being fetched is the last one
    const int k_next = kbeg + (s + 1) * BKT;
    if (more) {
      la.rotate();
      lb.rotate();
      la.load(p.A, p.lda, m0, k_next, a_span, tail, kbeg);
      lb.load(p.B, p.ldb, n0, k_next, b_span, tail, kbeg);
      la.prefetch_lines(p.a_idx, k_next + BKT, kend);
      lb.prefetch_lines(p.b_idx, k_next + BKT, kend);
    }
    // Groups of 4 MFMAs over 8 physical k: lane half h feeds k = 8g + 4h + q
    // to step q (any k <-> slot map is valid as long as A and B agree), so a
    // k-contiguous operand is ONE ds_read_b128 per 4 MFMAs.
    // (reading the operands of group g + 1 before issuing the MFMAs of group g was
    // measured: no gain -- 146.0 / 148.2 vs 145.2 / 144.7 ms -- the other waves of
    // the SIMD cover the LDS round trip already)
    auto read_frags = [&](float (&a)[TM][4], float (&b)[TN][4], int g) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(
              As + (wm0 + 32 * i + l31) * LDK + 8 * g + 4 * half);
          a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a[i][q] = As[(8 * g + 4 * half + q) * LDA_S + wm0 + 32 * i + l31];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(
              Bs + (wn0 + 32 * j + l31) * LDK + 8 * g + 4 * half);
          b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            b[j][q] = Bs[(8 * g + 4 * half + q) * LDB_S + wn0 + 32 * j + l31];
        }
      }
    };
    auto issue = [&](const float (&a)[TM][4], const float (&b)[TN][4]) {
      // the wave that is about to issue MFMAs goes ahead of co-resident waves that
      // are still loading / storing tiles (-0.8 % per C3 iteration, 2 x A/B)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q],
                                                             acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
#pragma unroll
    for (int g = 0; g < BKT / 8; ++g) {
      float a[TM][4], b[TN][4];
      read_frags(a, b, g);
      issue(a, b);
    }
    if (do_colsum) {
      const float* T = p.colsum_of_b ? Bs : As;
      const int LD = p.colsum_of_b ? LDB_S : LDA_S;
      const int W = p.colsum_of_b ? BN : BM;
      if ((int)threadIdx.x < W) {
#pragma unroll 8
        for (int k = 0; k < BKT; ++k) csum += T[k * LD + threadIdx.x];
      }
    }
    __syncthreads();
    if (more) {
      la.store(As, m0, p.M, k_next, kend, tail);
      lb.store(Bs, n0, p.N, k_next, kend, tail);
      __syncthreads();
    }
  }
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic (lgkmcnt) but not for
// global loads in flight (__syncthreads() also waits vmcnt(0): the prefetch of the tile
// after next would have to land first).
__device__ __forceinline__ void ga_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The k-loop for FEW waves per SIMD (256 x 256 tiles: 4 waves of 128 x 128, 256
// accumulator registers each, ONE wave per SIMD -- the shape the vendor library picks for
// the 65536 x 512 x 512 products of C5, profiles/r03_notes.md).  gemm_mainloop above relies
// on 4 waves per SIMD to hide its global -> LDS -> fragment latencies; with one wave the
// loop has to be software pipelined itself:
//   * two LDS stages, ONE barrier per k-step; tile s + 1 sits in registers while tile s
//     is multiplied and is written to the other stage behind the MFMAs of group G - 2,
//     after which the loads of tile s + 2 are issued (a whole k-step to land);
//   * the fragments of group g + 1 are read while group g's MFMAs run (two register
//     sets); the barrier sits in front of the LAST group, and the first fragments of the
//     next tile are read right behind it, under that group's MFMAs.
// Interior tiles only (FULL loaders; the last k-step may be partial).  Same k order per
// accumulator as gemm_mainloop: bit-identical results.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC, int BKT, int TM,
          int TN>
__device__ __forceinline__ void gemm_mainloop_pipe(const GemmParams& p, float* lds,
                                                   f32x16 (&acc)[TM][TN], int m0, int n0,
                                                   int kbeg, int kend, int wm0, int wn0) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD, LDK = BKT + PAD;
  constexpr int A_FLOATS = A_KC ? BM * LDK : BKT * LDA_S;
  constexpr int B_FLOATS = B_KC ? BN * LDK : BKT * LDB_S;
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  constexpr int G = BKT / 8;
  static_assert(G >= 2 && G % 2 == 0, "the fragment sets alternate per group");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, l31 = lane & 31;
  TileLoader<BM, A_KC, NT, BKT, true> la;
  TileLoader<BN, B_KC, NT, BKT, true> lb;
  const int a_span = A_KC ? p.K : p.M;
  const int b_span = B_KC ? p.K : p.N;
  const int nk = (kend - kbeg + BKT - 1) / BKT;
  if (nk <= 0) {
    __syncthreads();
    return;
  }
  la.init(p.a_idx, m0, p.M, kbeg, kend);
  lb.init(p.b_idx, n0, p.N, kbeg, kend);
  la.init_linear(p.A, p.lda, p.a_idx, m0, kbeg);
  lb.init_linear(p.B, p.ldb, p.b_idx, n0, kbeg);
  la.load(p.A, p.lda, m0, kbeg, a_span, nk == 1, kbeg);
  lb.load(p.B, p.ldb, n0, kbeg, b_span, nk == 1, kbeg);
  la.prefetch_lines(p.a_idx, kbeg + BKT, kend);
  lb.prefetch_lines(p.b_idx, kbeg + BKT, kend);
  la.store(lds, m0, p.M, kbeg, kend, nk == 1);
  lb.store(lds + A_FLOATS, n0, p.N, kbeg, kend, nk == 1);
  if (nk > 1) {
    la.rotate();
    lb.rotate();
    la.load(p.A, p.lda, m0, kbeg + BKT, a_span, nk == 2, kbeg);
    lb.load(p.B, p.ldb, n0, kbeg + BKT, b_span, nk == 2, kbeg);
    la.prefetch_lines(p.a_idx, kbeg + 2 * BKT, kend);
    lb.prefetch_lines(p.b_idx, kbeg + 2 * BKT, kend);
  }
  __syncthreads();

  float fa[2][TM][4], fb[2][TN][4];
  auto read_frags = [&](const float* As, const float* Bs, float (&a)[TM][4],
                        float (&b)[TN][4], int g) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (A_KC) {
        const float4 v = *reinterpret_cast<const float4*>(
            As + (wm0 + 32 * i + l31) * LDK + 8 * g + 4 * half);
        a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          a[i][q] = As[(8 * g + 4 * half + q) * LDA_S + wm0 + 32 * i + l31];
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(
            Bs + (wn0 + 32 * j + l31) * LDK + 8 * g + 4 * half);
        b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          b[j][q] = Bs[(8 * g + 4 * half + q) * LDB_S + wn0 + 32 * j + l31];
      }
    }
  };
  auto issue = [&](const float (&a)[TM][4], const float (&b)[TN][4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0,
                                                           0, 0);
  };
  read_frags(lds, lds + A_FLOATS, fa[0], fb[0], 0);

  // The loop body is ONE basic block (no tail masks: the caller guarantees whole k-steps;
  // the last step re-loads its own tile and stores it where nobody reads it), so that the
  // machine scheduler can be told how to interleave: a wave issues in order, and what is
  // to run beside the MFMAs has to sit BETWEEN them in the instruction stream.
  for (int s = 0; s < nk; ++s) {
    const float* As = lds + (s & 1) * STAGE;
    const float* Bs = As + A_FLOATS;
    float* An = lds + ((s + 1) & 1) * STAGE;
    float* Bn = An + A_FLOATS;
    const int k1 = kbeg + min(s + 1, nk - 1) * BKT;  // tile in the registers
    const int k2 = kbeg + min(s + 2, nk - 1) * BKT;  // tile to fetch
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) {
        read_frags(As, Bs, fa[(g + 1) & 1], fb[(g + 1) & 1], g + 1);
      } else {
        ga_lds_barrier();
        read_frags(An, Bn, fa[0], fb[0], 0);
      }
      if (g == G - 2) {
        la.store(An, m0, p.M, k1, kend, false);
        lb.store(Bn, n0, p.N, k1, kend, false);
        la.rotate();
        lb.rotate();
        la.load(p.A, p.lda, m0, k2, a_span, false, kbeg);
        lb.load(p.B, p.ldb, n0, k2, b_span, false, kbeg);
        la.prefetch_lines(p.a_idx, k2 + BKT, kend);
        lb.prefetch_lines(p.b_idx, k2 + BKT, kend);
      }
      issue(fa[g & 1], fb[g & 1]);
      if (g == G - 2) {
        // 4 MFMAs, then one LDS store and one global load, 16 times
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
  }
  __syncthreads();  // (the epilogue stages output rows over the operand stages)
}

}  // namespace
