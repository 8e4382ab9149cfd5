// fp32 MFMA GEMM family for the MLP policy / value networks (gfx950).
//
// One LDS-tiled kernel, C(m,n) = epi(sum_k A(m,k) * B(k,n)), instantiated for
// the three products an MLP layer needs (reference: nn.Linear forward/backward
// inside torch/modules/multi_headed_mlp_module.py:136-151, driven by
// torch/algos/vpg.py:250-293):
//   forward      Y  = act(X W^T + b)          A, B both k-contiguous
//   data grad    dX = (dY W) * (1 - H^2)      A k-contiguous, B n-contiguous
//   weight grad  dW = dY^T X (+ db = 1^T dY)  A, B both "row"-contiguous,
//                                             split-K over the batch into slabs
// Arithmetic is exact fp32 on v_mfma_f32_32x32x2_f32 (the parity bar is 1e-5 on
// losses, so no reduced-precision operands).  Every matrix this file touches
// has a leading dimension that is a multiple of 4 floats and a 16-B aligned
// base (garage_amd pads obs / weights / activations accordingly), so all
// global traffic is 16-B vector loads; ragged edges are masked, not branched.
//
// Operand tiles keep their memory orientation in LDS.  A k-contiguous operand is
// stored [BR][BK + 4] (one ds_write_b128 per loaded vector) and feeds four MFMAs
// from one ds_read_b128: lane l of a 32x32x2 MFMA holds row l & 31 and, in MFMA q
// of group g, k = 8g + 4(l >> 5) + q -- any k <-> slot map is valid as long as A
// and B agree.  A row-contiguous operand is stored [BK][BR + 4] and read with one
// ds_read_b32 per MFMA under the same map.  The +4 floats keep rows 16-B aligned
// and the 32-lane reads of an instruction on distinct banks (SQ_LDS_BANK_CONFLICT
// = 0 by PMC).  Tile shapes: 128x128 (8 waves, two workgroups per CU), 64x64 for
// 33..64-wide outputs, 128x32 for narrow ones; the products with a dimension
// <= 32 that stream a whole activation matrix go to skinny.hip instead.
#include "common.h"
#include <hip/hip_ext.h>

#include "prof.h"

#include "gemm_core.h"
#include "fused_train.h"

namespace {

// The narrow layer on 64 staged rows: lane = row, wave = a segment of CPS columns,
// whose weights are wave-uniform and come through the scalar cache (constant
// address space), so the inner product is v_fmac with an SGPR operand and LDS is
// read once per element; the waves' partial sums meet in `hp` and are added in a
// fixed order.
template <int NT_ALL, int BN, int LDC>
__device__ __forceinline__ void head_on_staged_rows(const GemmParams& p,
                                                    const float* stage, float* hp,
                                                    int n0, int m_base) {
  typedef const __attribute__((address_space(4))) float* uniform_ptr;
  constexpr int SEGS = NT_ALL / 64, CPS = BN / SEGS, HN = 8;
  const int row = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4 h[CPS / 4];
#pragma unroll
  for (int i = 0; i < CPS / 4; ++i)
    h[i] = *reinterpret_cast<const float4*>(stage + row * LDC + seg * CPS + 4 * i);
  float a[HN];
#pragma unroll
  for (int j = 0; j < HN; ++j) {
    a[j] = 0.f;
    if (j < p.head_n) {
      uniform_ptr w = (uniform_ptr)(uintptr_t)(p.head_W + (int64_t)j * p.head_ldw + n0 +
                                               seg * CPS);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < CPS / 4; ++i) {
        s0 = fmaf(h[i].x, w[4 * i + 0], s0);
        s1 = fmaf(h[i].y, w[4 * i + 1], s1);
        s0 = fmaf(h[i].z, w[4 * i + 2], s0);
        s1 = fmaf(h[i].w, w[4 * i + 3], s1);
      }
      a[j] = s0 + s1;
    }
  }
  // hp holds PL planes: waves PL.. add theirs onto plane (wave - PL) in a second step
  constexpr int PL = SEGS < 4 ? SEGS : 4;
  static_assert(SEGS <= 2 * PL, "two reduction steps");
  float* mine = hp + ((seg % PL) * 64 + row) * HN;
  if (seg < PL) {
    *reinterpret_cast<float4*>(mine) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(mine + 4) = make_float4(a[4], a[5], a[6], a[7]);
  }
  __syncthreads();
  if (SEGS > PL) {
    if (seg >= PL) {
      float4 lo = *reinterpret_cast<const float4*>(mine);
      float4 hi = *reinterpret_cast<const float4*>(mine + 4);
      lo.x += a[0]; lo.y += a[1]; lo.z += a[2]; lo.w += a[3];
      hi.x += a[4]; hi.y += a[5]; hi.z += a[6]; hi.w += a[7];
      *reinterpret_cast<float4*>(mine) = lo;
      *reinterpret_cast<float4*>(mine + 4) = hi;
    }
    __syncthreads();
  }
  for (int o = threadIdx.x; o < 64 * HN; o += NT_ALL) {
    const int r = o / HN, j = o % HN;
    if (j < p.head_n) {
      float s = p.head_bias ? p.head_bias[j] : 0.f;
#pragma unroll
      for (int w = 0; w < PL; ++w) s += hp[(w * 64 + r) * HN + j];
      p.head_out[(int64_t)(m_base + r) * p.head_ld + j] = s;
    }
  }
}

// Split-operand k-loop of a 128 x 128 tile (opt-in, gemm_core.h: ft_split3_pair): the A
// operand -- k-contiguous rows of activations -- is fetched as two 16-B quads per thread
// and 32-deep step, two steps ahead, split into three bf16 planes in LDS one step ahead
// (two buffers); the B operand -- a weight matrix -- comes as fragments straight from
// its planes in L2 (p.bplanes), one step ahead, refilled in place; 24 MFMAs per step
// and wave (v_mfma_f32_32x32x16_bf16), one barrier.  K % 4 == 0; rows beyond M and k
// beyond K are zero in the planes.
__device__ __forceinline__ void gemm_mainloop_split(const GemmParams& p, float* lds,
                                                    f32x16 (&acc)[2][1], int m0, int n0,
                                                    int wm0, int wn0) {
  constexpr int PLANE_B = 128 * FT_PLANE_ROW_B, ABUF_B = 3 * PLANE_B;
  char* apl = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int K = p.K, nk = (K + 31) / 32;
  const int qq = tid & 7;
  const float* arow[2];
  bool row_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (tid >> 3) + 64 * i;
    const int m = min(m0 + r, p.M - 1);
    const int64_t line = p.a_idx ? (int64_t)p.a_idx[m] : (int64_t)m;
    arow[i] = p.A + line * p.lda;
    row_ok[i] = m0 + r < p.M;
  }
  auto load_quads = [&](int s, float4 (&v)[2]) {
    const int kq = min(32 * s + 4 * qq, K - 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const float4*>(arow[i] + kq);
  };
  auto store_quads = [&](int s, const float4 (&v)[2]) {
    const bool k_ok = 32 * s + 4 * qq < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 x = v[i];
      if (!row_ok[i] || !k_ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t hi[2], mid[2], lo[2];
      ft_split3_pair(x.x, x.y, hi[0], mid[0], lo[0]);
      ft_split3_pair(x.z, x.w, hi[1], mid[1], lo[1]);
      char* dst = apl + (s & 1) * ABUF_B + ((tid >> 3) + 64 * i) * FT_PLANE_ROW_B + qq * 8;
      *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0], hi[1]);
      *reinterpret_cast<uint2*>(dst + PLANE_B) = make_uint2(mid[0], mid[1]);
      *reinterpret_cast<uint2*>(dst + 2 * PLANE_B) = make_uint2(lo[0], lo[1]);
    }
  };
  const uint16_t* wpl = p.bplanes + (((n0 + wn0) / 32) * 64 + lane) * 8;
  ft_u32x4 bw[2][3];
  auto fetch_planes = [&](int s, int g) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      bw[g][pl] = *reinterpret_cast<const ft_u32x4*>(
          wpl + pl * p.bplane_stride + (int64_t)((2 * s + g) * p.bplane_nblk) * 512);
  };
  if (nk <= 0) return;
  fetch_planes(0, 0);
  fetch_planes(0, 1);
  float4 qc[2], qn[2];
  load_quads(0, qc);
  if (nk > 1) load_quads(1, qn);
  store_quads(0, qc);
  qc[0] = qn[0]; qc[1] = qn[1];
  __syncthreads();
#define GS_SB __builtin_amdgcn_sched_barrier(0)
  for (int s = 0; s < nk; ++s) {
    const bool more = s + 1 < nk;
    const char* arow_l =
        apl + (s & 1) * ABUF_B + (wm0 + l31) * FT_PLANE_ROW_B + 16 * half;
    ft_u32x4 af[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        af[0][i][pl] = *reinterpret_cast<const ft_u32x4*>(arow_l + pl * PLANE_B +
                                                          32 * i * FT_PLANE_ROW_B);
#ifndef GA_GABL_NOLOAD
    if (s + 2 < nk) load_quads(s + 2, qn);
#endif
    GS_SB;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int slot = 0; slot < 24; ++slot) {
      const int g = slot / 12, t = (slot % 12) / 2, i = slot % 2;
      const int pa = t == 0 ? 2 : (t == 1 || t == 2) ? 1 : 0;
      const int pb = t == 3 ? 2 : (t == 1 || t == 4) ? 1 : 0;
#ifdef GA_GABL_NOMFMA
      acc[i][0][slot & 15] += __uint_as_float(af[g][i][pa][0] ^ bw[g][pb][0]);
#else
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(ft_bf16x8, af[g][i][pa]),
          __builtin_bit_cast(ft_bf16x8, bw[g][pb]), acc[i][0], 0, 0, 0);
#endif
      GS_SB;
      if (slot == 6 || slot == 8 || slot == 11) {
        const int pl = slot == 6 ? 2 : slot == 8 ? 1 : 0;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
          af[1][ii][pl] = *reinterpret_cast<const ft_u32x4*>(
              arow_l + pl * PLANE_B + 32 * ii * FT_PLANE_ROW_B + 32);
      }
      if (more) {
#ifndef GA_GABL_NOSTORE
        if (slot == 2) store_quads(s + 1, qc);
#endif
#ifndef GA_GABL_NOFETCH
        if (slot == 11) fetch_planes(s + 1, 0);
        if (slot == 23) fetch_planes(s + 1, 1);
#endif
      }
      GS_SB;
    }
    __builtin_amdgcn_s_setprio(0);
    qc[0] = qn[0]; qc[1] = qn[1];
    __syncthreads();
  }
#undef GS_SB
}

// bid: the workgroup's linear id within ITS problem (a pair launch carries two)
// SPLIT: the opt-in split-operand k-loop above (128 x 128 tiles, A_KC, weights as planes)
// PIPE: interior tiles take the software-pipelined two-stage k-loop
// (gemm_core.h: gemm_mainloop_pipe) -- for tile shapes with few waves per SIMD
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC, int BKT,
          bool HEAD, bool SPLIT = false, bool PIPE = false>
__device__ __forceinline__ void gemm_f32_body(const GemmParams& p, const int bid) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  // row-contiguous operands: [BK][BR + PAD]; k-contiguous ones: [BR][BK + PAD]
  constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD, LDK = BKT + PAD;
  constexpr int A_FLOATS = A_KC ? BM * LDK : BKT * LDA_S;
  constexpr int B_FLOATS = B_KC ? BN * LDK : BKT * LDB_S;
  // (the epilogue stages 64 output rows in the same memory)
  constexpr int STAGE_FLOATS = (BM % 64 == 0) ? 64 * (BN + 4) : 0;
  constexpr int OPERAND_FLOATS = (PIPE ? 2 : 1) * (A_FLOATS + B_FLOATS);
  constexpr int TILE_FLOATS = OPERAND_FLOATS > STAGE_FLOATS ? OPERAND_FLOATS : STAGE_FLOATS;
  // HEAD: + the waves' partial head sums, [<= 4 planes][64 rows][8]
  constexpr int HEAD_PLANES = WAVES_M * WAVES_N < 4 ? WAVES_M * WAVES_N : 4;
  // (SPLIT: two buffers of three bf16 planes of the 128-row A tile)
  constexpr int SPLIT_FLOATS = SPLIT ? 2 * 3 * 128 * FT_PLANE_ROW_B / 4 : 0;
  constexpr int LDS_FLOATS = (TILE_FLOATS > SPLIT_FLOATS ? TILE_FLOATS : SPLIT_FLOATS) +
                             (HEAD ? HEAD_PLANES * 64 * 8 : 0);
  static_assert(!SPLIT || (BM == 128 && BN == 128 && WAVES_M == 2 && WAVES_N == 4 && A_KC &&
                           !HEAD),
                "the split-operand loop: 128 x 128 tiles, 8 waves, k-contiguous A");
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;
  // logical block from the linear id (XCD-aware order, common.h): with one split
  // the n blocks of an m block share its A tile; with split-K every block of a
  // split shares the split's A and B rows
  int bx, by, bz;
  if (p.gz == 1) {
    ga_xcd_group(bid, p.gx, p.gy, &bx, &by);
    bz = 0;
  } else {
    int mem;
    ga_xcd_group(bid, p.gz, p.gx * p.gy, &bz, &mem);
    bx = mem % p.gx;
    by = mem / p.gx;
  }
  const int m0 = bx * BM;
  const int n0 = by * BN;
  const int split = bz;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float csum = 0.f;  // colsum accumulator (threads < BM or < BN)
  const bool do_colsum =
      p.colsum != nullptr &&
      (p.colsum_of_b ? (bx == 0) : (by == 0));

  // interior workgroups (every one at the C3 shapes, all but the last row / column
  // block and the last split otherwise) take the mask-free loader
  const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N) && kend > kbeg;
  if constexpr (SPLIT) {
    gemm_mainloop_split(p, lds, acc, m0, n0, wm0, wn0);
    __syncthreads();  // (the epilogue stages output rows over the planes)
  } else if (PIPE && full && !do_colsum)
    gemm_mainloop_pipe<BM, BN, WAVES_M, WAVES_N, A_KC, B_KC, BKT>(p, lds, acc, m0, n0, kbeg,
                                                                  kend, wm0, wn0);
  else if (full)
    gemm_mainloop<BM, BN, WAVES_M, WAVES_N, A_KC, B_KC, BKT, true>(
        p, lds, acc, csum, do_colsum, m0, n0, kbeg, kend, wm0, wn0);
  else
    gemm_mainloop<BM, BN, WAVES_M, WAVES_N, A_KC, B_KC, BKT, false>(
        p, lds, acc, csum, do_colsum, m0, n0, kbeg, kend, wm0, wn0);

  // ---- epilogue: D(row, col): col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cout = p.C + (int64_t)split * p.c_split_stride;
#ifndef GA_NO_TR_EPILOGUE
  // Interior tiles of a row-major C go through LDS, 64 rows at a time in the
  // operand stages that are free now: a lane then owns 4
  // adjacent columns of a row, so stores (and the H / C loads of the fused
  // epilogues) are 16-B accesses and a wave instruction covers two whole 512-B
  // row segments instead of two 128-B ones.
  constexpr int NT_ALL = 64 * WAVES_M * WAVES_N;
  constexpr bool TR_OK = BM % 64 == 0 && (64 * (BN / 4)) % NT_ALL == 0;
  // (PIPE: the host has checked the layout conditions -- gemm_pipe_ok -- and N is a
  // multiple of BN; the last row block may be ragged and guards its rows below.  The
  // per-element path at the end is not instantiated: its loops stay rolled at 16
  // accumulator tiles and the accumulators would then live in scratch memory.)
  if (PIPE || (TR_OK && full && p.c_cs == 1 && (p.c_rs & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(Cout) & 15u) == 0 &&
      (!p.H || ((p.ldh & 3) == 0 && (reinterpret_cast<uintptr_t>(p.H) & 15u) == 0)) &&
      (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15u) == 0))) {
    constexpr int LDC = BN + 4;
    for (int hrow = 0; hrow < BM; hrow += 64) {
      // (a wave's rows may span more than one 64-row pass: WM = 128 in the
      // one-wave-per-SIMD A/B instantiation)
      if (wm0 < hrow + 64 && wm0 + WM > hrow) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rr =
                  wm0 - hrow + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              if (rr >= 0 && rr < 64)
                lds[rr * LDC + wn0 + 32 * j + (lane & 31)] = acc[i][j][r];
            }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < (TR_OK ? 64 * (BN / 4) / NT_ALL : 0); ++q) {
        const int idx = threadIdx.x + NT_ALL * q;
        const int rr = idx / (BN / 4), c4 = idx % (BN / 4);
        const int m = m0 + hrow + rr, n = n0 + 4 * c4;
        if (PIPE && m >= p.M) continue;
        float4 v = *reinterpret_cast<const float4*>(lds + rr * LDC + 4 * c4);
        float* dst = Cout + (int64_t)m * p.c_rs + n;
        if (p.accum) {
          const float4 o = *reinterpret_cast<const float4*>(dst);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (p.epi == EPI_BIAS_ACT) {
          if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          if (p.act) {
            v.x = act_apply(v.x, p.act); v.y = act_apply(v.y, p.act);
            v.z = act_apply(v.z, p.act); v.w = act_apply(v.w, p.act);
          }
        }
        if ((p.epi == EPI_BIAS_ACT && p.H) || p.epi == EPI_MUL_DTANH) {
          const float4 h =
              *reinterpret_cast<const float4*>(p.H + (int64_t)m * p.ldh + n);
          v.x *= act_slope(h.x, p.hact); v.y *= act_slope(h.y, p.hact);
          v.z *= act_slope(h.z, p.hact); v.w *= act_slope(h.w, p.hact);
        }
        *reinterpret_cast<float4*>(dst) = v;
        if constexpr (HEAD) *reinterpret_cast<float4*>(lds + rr * LDC + 4 * c4) = v;
      }
      __syncthreads();
      if constexpr (HEAD) {
        head_on_staged_rows<NT_ALL, BN, LDC>(p, lds, lds + TILE_FLOATS, n0, m0 + hrow);
        __syncthreads();
      }
    }
    if (do_colsum) {
      const int W = p.colsum_of_b ? BN : BM;
      const int base = p.colsum_of_b ? n0 : m0;
      const int lim = p.colsum_of_b ? p.N : p.M;
      if ((int)threadIdx.x < W && base + (int)threadIdx.x < lim)
        p.colsum[(int64_t)split * p.colsum_split_stride + base + threadIdx.x] = csum;
    }
    return;
  }
#endif
  if constexpr (!PIPE) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn0 + 32 * j + (lane & 31);
      if (n >= p.N) continue;
      float bias = 0.f;
      if (p.epi == EPI_BIAS_ACT && p.bias) bias = p.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (p.accum) v += Cout[(int64_t)m * p.c_rs + (int64_t)n * p.c_cs];
        if (p.epi == EPI_BIAS_ACT) {
          v += bias;
          v = act_apply(v, p.act);
          if (p.H) {
            const float h = p.H[(int64_t)m * p.ldh + n];
            v *= act_slope(h, p.hact);
          }
        } else if (p.epi == EPI_MUL_DTANH) {
          const float h = p.H[(int64_t)m * p.ldh + n];
          v *= act_slope(h, p.hact);
        }
        Cout[(int64_t)m * p.c_rs + (int64_t)n * p.c_cs] = v;
      }
    }
  }
  if (do_colsum) {
    const int W = p.colsum_of_b ? BN : BM;
    const int base = p.colsum_of_b ? n0 : m0;
    const int lim = p.colsum_of_b ? p.N : p.M;
    if ((int)threadIdx.x < W && base + (int)threadIdx.x < lim)
      p.colsum[(int64_t)split * p.colsum_split_stride + base + threadIdx.x] = csum;
  }
  }  // !PIPE
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC,
          int BKT = BK, bool HEAD = false>
#ifdef GA_GEMM_WAVES_PER_EU  // A/B builds (tools/build_variants.sh)
__attribute__((amdgpu_waves_per_eu(GA_GEMM_WAVES_PER_EU, 8)))
#endif
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_f32_kernel(
    GemmParams p) {
  gemm_f32_body<BM, BN, WAVES_M, WAVES_N, A_KC, B_KC, BKT, HEAD>(p, (int)blockIdx.x);
}

// 256 x 256 tiles, 4 waves of 128 x 128: one wave per SIMD, pipelined k-loop
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_pipe_kernel(GemmParams p) {
  gemm_f32_body<256, 256, 2, 2, A_KC, B_KC, BK, false, false, true>(p, (int)blockIdx.x);
}

__global__ __launch_bounds__(512, 4) void gemm_kc_split_kernel(GemmParams p) {
  gemm_f32_body<128, 128, 2, 4, true, true, BK, false, true>(p, (int)blockIdx.x);
}

// Two problems of the same shape in one grid (the policy's and the value function's
// weight-gradient GEMM of one optimizer step): workgroup b takes workgroup b / 2 of
// problem b % 2 (see fwd_head_loss_pair_kernel, fused_train.hip).
struct GemmPair {
  GemmParams a, b;
};
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_f32_pair_kernel(
    GemmPair pp) {
  const int bid = (int)(blockIdx.x >> 1);
  if (blockIdx.x & 1)
    gemm_f32_body<BM, BN, WAVES_M, WAVES_N, A_KC, B_KC, BK, false>(pp.b, bid);
  else
    gemm_f32_body<BM, BN, WAVES_M, WAVES_N, A_KC, B_KC, BK, false>(pp.a, bid);
}

// ---- split-operand weight-gradient GEMM (opt-in, see ft_split3 in gemm_core.h):
// C[m][n] = sum_k A(m, k) B(k, n) with BOTH operands row-contiguous in memory
// (A(m, k) = A[k * lda + m]: k is the minibatch row).  v_mfma_f32_32x32x16_bf16 wants
// 8 consecutive k per lane, so the loader pairs two consecutive rows: a thread loads
// the same 16-B column quad of rows 2 t and 2 t + 1, splits the eight values and
// writes, per plane, the four (row 2 t, row 2 t + 1) bf16 pairs as ONE 16-B LDS store
// into [pair][column] dwords; a fragment is then four ds_read_b32 down the pairs.
// 128 x 128 tiles, 8 waves (2 x 4, 64 x 32 each), 16 rows per step, operands fetched
// three steps ahead (they come from HBM), double-buffered planes, one barrier per
// step.  Interior shapes only (the host checks): M, N multiples of 128, no gathers.
constexpr int NTS_LD = 136;                  // dwords per pair row: 128 + 8 (the two lane
                                             // halves read rows 4 apart: 32 banks apart)
constexpr int NTS_PLANE = 8 * NTS_LD;        // dwords per plane of a 16-row step
constexpr int NTS_OPER = 3 * NTS_PLANE;      // hi, mid, lo
constexpr int NTS_BUF = 2 * NTS_OPER;        // A and B

__global__ __launch_bounds__(512, 4) void gemm_nt_split_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * NTS_BUF];
  __shared__ __attribute__((aligned(16))) float cs_lds[8 * 128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / 4) * 64, wn0 = (wave % 4) * 32;
  int bz, mem;
  ga_xcd_group((int)blockIdx.x, p.gz, p.gx * p.gy, &bz, &mem);
  const int bx = mem % p.gx, by = mem / p.gx;
  const int m0 = bx * 128, n0 = by * 128;
  const int kbeg = bz * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + 15) / 16;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // loader: threads 0 .. 255 take A, 256 .. 511 take B; pair pi of the step, quad c4
  const bool isb = tid >= 256;
  const int lt = tid & 255, pi = lt >> 5, c4 = lt & 31;
  const float* src = isb ? p.B + n0 + 4 * c4 : p.A + m0 + 4 * c4;
  const int64_t ld = isb ? p.ldb : p.lda;
  uint32_t* dst0 = lds + (isb ? NTS_OPER : 0) + pi * NTS_LD + 4 * c4;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load2 = [&](int s, float4& v0, float4& v1) {
    const int r0 = kbeg + 16 * s + 2 * pi;
    v0 = *reinterpret_cast<const float4*>(src + (int64_t)min(r0, p.K - 1) * ld);
    v1 = *reinterpret_cast<const float4*>(src + (int64_t)min(r0 + 1, p.K - 1) * ld);
    if (r0 >= kend) v0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + 1 >= kend) v1 = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store2 = [&](int s, const float4& v0, const float4& v1) {
    const float a[4] = {v0.x, v0.y, v0.z, v0.w}, b[4] = {v1.x, v1.y, v1.z, v1.w};
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ft_split3_pair(a[j], b[j], h[j], m[j], l[j]);
    uint32_t* d = dst0 + (s & 1) * NTS_BUF;
    *reinterpret_cast<uint4*>(d) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(d + NTS_PLANE) = make_uint4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<uint4*>(d + 2 * NTS_PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
    if (!isb) {
      csum.x += v0.x + v1.x; csum.y += v0.y + v1.y;
      csum.z += v0.z + v1.z; csum.w += v0.w + v1.w;
    }
  };
  // three steps in flight
  float4 q0[3], q1[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    q0[d] = make_float4(0.f, 0.f, 0.f, 0.f);
    q1[d] = q0[d];
    if (d < nk) load2(d, q0[d], q1[d]);
  }
  if (nk > 0) store2(0, q0[0], q1[0]);
  __syncthreads();

  const int half = lane >> 5, l31 = lane & 31;
  // fragment rows: pairs 4 half .. 4 half + 3 of the step
  const uint32_t* fa = lds + 4 * half * NTS_LD + wm0 + l31;
  const uint32_t* fb = lds + NTS_OPER + 4 * half * NTS_LD + wn0 + l31;
  auto frag = [&](const uint32_t* base, int pl) {
    ft_u32x4 v;
    v[0] = base[pl * NTS_PLANE];
    v[1] = base[pl * NTS_PLANE + NTS_LD];
    v[2] = base[pl * NTS_PLANE + 2 * NTS_LD];
    v[3] = base[pl * NTS_PLANE + 3 * NTS_LD];
    return __builtin_bit_cast(ft_bf16x8, v);
  };
#pragma unroll 1
  for (int s0 = 0; s0 < nk; s0 += 3) {
    // (unrolled by the depth of the prefetch ring so that its registers are static)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int s = s0 + d;
      if (s < nk) {
        const uint32_t* ab = fa + (s & 1) * NTS_BUF;
        const uint32_t* bb = fb + (s & 1) * NTS_BUF;
        ft_bf16x8 a[2][3], b[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          a[0][pl] = frag(ab, pl);
          a[1][pl] = frag(ab + 32, pl);
          b[pl] = frag(bb, pl);
        }
        // the quads of step s + 1 are in registers since two steps ago: split them
        // into the other buffer while this step's MFMAs run, then refill the slot
        // with step s + 3
        __builtin_amdgcn_s_setprio(1);
        ft_mfma6(a[0], b, acc[0]);
        ft_mfma6(a[1], b, acc[1]);
        __builtin_amdgcn_s_setprio(0);
        if (s + 1 < nk) store2(s + 1, q0[(d + 1) % 3], q1[(d + 1) % 3]);
        if (s + 3 < nk) load2(s + 3, q0[d], q1[d]);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: the slab of this split; bias-gradient column sums of A
  float* Cout = p.C + (int64_t)bz * p.c_split_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
      Cout[(int64_t)m * p.c_rs + (int64_t)(n0 + wn0 + l31) * p.c_cs] = acc[i][r];
    }
  if (p.colsum != nullptr && by == 0) {
    if (!isb) *reinterpret_cast<float4*>(cs_lds + pi * 128 + 4 * c4) = csum;
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += cs_lds[k * 128 + tid];
      p.colsum[(int64_t)bz * p.colsum_split_stride + m0 + tid] = t;
    }
  }
}

static int g_small_m = 1;

template <bool A_KC, bool B_KC>
int launch_gemm(const GemmParams& p_in, int splits, hipStream_t stream) {
  GemmParams p = p_in;
  const int mode = (A_KC && B_KC) ? 0 : (A_KC ? 1 : 2);
  // algorithmic flops: 2 M N K (the padding of ragged tiles is not counted)
  const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_small_m && splits == 1 && p.K > 64 &&
      (p.N > 32 ? ga_ceil_div(p.M, 128) * ga_ceil_div(p.N, 128) <= 128 : p.M <= 4096)) {
    // fewer 128x128 tiles than half the CUs: the k-loop of one workgroup is then a
    // chain of dependent memory round trips (the weights were rewritten by the
    // optimizer step just before, so they come from beyond this XCD's L2) with
    // nothing else on the CU to hide them.  64x64 tiles spread rows and columns
    // over 4x the workgroups and a 128-deep k-step cuts the chain to K / 128 round
    // trips with 16 16-B loads in flight per thread.  Same k order per output
    // element as every other tile shape, so the results are bit-identical.
    // MLP(256,256) forward + backward at 64 rows: 43.6 + 54.3 -> 26.5 + 33.3 us,
    // at 4096 rows 43.5 + 81.2 -> 31.3 + 66.8 us; MLP(512,512,512) at 64 rows
    // 149.5 + 141.9 -> 61.7 + 69.5 us (tools/small_m_ab.py).  Narrow outputs (the
    // head layer) take the same kernel up to 4096 rows: a half-empty tile costs
    // nothing when the launch is one round-trip chain anyway.
    p.gx = (int)ga_ceil_div(p.M, 64); p.gy = (int)ga_ceil_div(p.N, 64); p.gz = 1;
    p.k_per_split = (int)ga_ceil_div(p.K, 128) * 128;
    dim3 grid((unsigned)(p.gx * p.gy));
    ga_prof_events(GA_PROF_GEMM_NT_128 + mode, flops, &e0, &e1);
    hipExtLaunchKernelGGL((gemm_f32_kernel<64, 64, 2, 2, A_KC, B_KC, 128>), grid,
                          dim3(256), 0, stream, e0, e1, 0, p);
  } else if (p.N <= 32) {
    // narrow outputs are HBM bound on the wide operand: 128-row tiles give
    // M/128 workgroups (256 at the C3 minibatch) to pull it through
    p.gx = (int)ga_ceil_div(p.M, 128); p.gy = (int)ga_ceil_div(p.N, 32); p.gz = splits;
    dim3 grid((unsigned)(p.gx * p.gy * p.gz));
    ga_prof_events(GA_PROF_GEMM_NT_256 + mode, flops, &e0, &e1);
    // (a 128-deep k tile -- 80 KB in flight per workgroup -- was tried for these
    // latency-bound shapes and measured no better than 32: one workgroup per CU)
    hipExtLaunchKernelGGL((gemm_f32_kernel<128, 32, 4, 1, A_KC, B_KC>), grid,
                          dim3(256), 0, stream, e0, e1, 0, p);
  } else if (p.N <= 64) {
    // 33..64 wide outputs (the 64-unit layers of the CartPole-shaped config): a
    // 128x128 tile would be half empty and give M/128 workgroups; 64x64 tiles
    // (4 waves, one 32x32 accumulator each) give 4x as many
    p.gx = (int)ga_ceil_div(p.M, 64); p.gy = 1; p.gz = splits;
    dim3 grid((unsigned)(p.gx * p.gy * p.gz));
    ga_prof_events(GA_PROF_GEMM_NT_128 + mode, flops, &e0, &e1);
    hipExtLaunchKernelGGL((gemm_f32_kernel<64, 64, 2, 2, A_KC, B_KC>), grid, dim3(256),
                          0, stream, e0, e1, 0, p);
  } else if (A_KC && ga_split_bf16_gemm() && splits == 1 && p.N >= 128 && p.K >= 128 &&
             p.K % 4 == 0 && p.M >= 1024 && !p.b_idx && !p.accum && !p.colsum &&
             !p.head_W && p.c_cs == 1) {
    // wide layers: forward (B(k, n) = W[n][k]) and data gradient (B(k, n) = W[k][n]);
    // B_KC tells which orientation the weight planes are needed in
    p.bplanes = B_KC ? ga_weight_planes(p.B, p.ldb, p.N, p.K, 0, stream)
                     : ga_weight_planes(p.B, p.ldb, p.K, p.N, 1, stream);
    GA_REQUIRE(p.bplanes, "gemm: no memory for the weight planes");
    const int np = (p.N + 31) & ~31, kp = (p.K + 31) & ~31;
    p.bplane_stride = (int64_t)np * kp;
    p.bplane_nblk = np / 32;
    p.gx = (int)ga_ceil_div(p.M, 128); p.gy = (int)ga_ceil_div(p.N, 128); p.gz = 1;
    p.k_per_split = (int)ga_ceil_div(p.K, BK) * BK;
    dim3 grid((unsigned)(p.gx * p.gy));
    ga_prof_events(GA_PROF_GEMM_NT_128 + mode, flops, &e0, &e1);
    hipExtLaunchKernelGGL(gemm_kc_split_kernel, grid, dim3(512), 0, stream, e0, e1, 0, p);
  } else if (!A_KC && !B_KC && ga_split_bf16_enabled() && p.M % 128 == 0 &&
             p.N % 128 == 0 && !p.a_idx && !p.b_idx && p.epi == EPI_PLAIN && !p.accum &&
             p.k_per_split % 16 == 0 && !p.colsum_of_b) {
    p.gx = p.M / 128; p.gy = p.N / 128; p.gz = splits;
    dim3 grid((unsigned)(p.gx * p.gy * p.gz));
    ga_prof_events(GA_PROF_GEMM_NT_128 + mode, flops, &e0, &e1);
    hipExtLaunchKernelGGL(gemm_nt_split_kernel, grid, dim3(512), 0, stream, e0, e1, 0, p);
  } else {
    p.gx = (int)ga_ceil_div(p.M, 128); p.gy = (int)ga_ceil_div(p.N, 128); p.gz = splits;
    dim3 grid((unsigned)(p.gx * p.gy * p.gz));
    ga_prof_events(GA_PROF_GEMM_NT_128 + mode, flops, &e0, &e1);
    // 8 waves (64x32 each): two workgroups per CU put 4 waves on every SIMD, so
    // the matrix pipe has work while other waves sit at the barrier / vmcnt
#ifdef GA_GEMM_BIG_TILE  // A/B builds: 4-wave workgroups with 128-row wave tiles
    // 1: 256 x 256 tiles, waves of 128 x 128 (one per SIMD); 2: 256 x 128, waves of 128 x 64
    if (splits == 1 && p.M >= 4096 && p.N % 256 == 0) {
      constexpr int TBN = GA_GEMM_BIG_TILE == 2 ? 128 : 256;
      p.gx = (int)ga_ceil_div(p.M, 256); p.gy = p.N / TBN; p.gz = 1;
      dim3 g2((unsigned)(p.gx * p.gy));
      const bool pipe_ok = !p.colsum && p.c_cs == 1 && (p.c_rs & 3) == 0 && ga_aligned16(p.C) &&
                           (!p.H || ((p.ldh & 3) == 0 && ga_aligned16(p.H))) &&
                           (!p.bias || ga_aligned16(p.bias)) && !p.accum &&
                           p.K % BK == 0 && !p.a_idx && !p.b_idx;
      if (GA_GEMM_BIG_TILE == 3 && pipe_ok)
        hipExtLaunchKernelGGL((gemm_f32_pipe_kernel<A_KC, B_KC>), g2, dim3(256), 0, stream,
                              e0, e1, 0, p);
      else
        hipExtLaunchKernelGGL((gemm_f32_kernel<256, TBN, 2, 2, A_KC, B_KC>), g2, dim3(256),
                              0, stream, e0, e1, 0, p);
      GA_CHECK_LAUNCH("gemm_f32 (256-row tiles)");
      return GA_OK;
    }
#endif
#ifdef GA_GEMM_BIG_BKT  // A/B builds (tools/build_variants.sh): k-tile depth of this shape
    hipExtLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 4, A_KC, B_KC, GA_GEMM_BIG_BKT>),
                          grid, dim3(512), 0, stream, e0, e1, 0, p);
#else
    hipExtLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 4, A_KC, B_KC>), grid,
                          dim3(512), 0, stream, e0, e1, 0, p);
#endif
  }
  GA_CHECK_LAUNCH("gemm_f32");
  return GA_OK;
}

inline int round4(int v) { return (v + 3) & ~3; }

// The last hidden layer and the head layer in one launch (tiles that span the
// layer's whole width: 64, 128 or 256 units).  Returns 1 when the shape is not
// taken -- every workgroup must be on the staged-epilogue path, which the kernel
// decides from the same conditions.
int launch_gemm_with_head(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  const int bm = 64;
  if (!(p.N == 64 || p.N == 128 || p.N == 256) || p.M % bm != 0 || p.M < bm ||
      p.head_n < 1 || p.head_n > 8 || p.head_ld < p.head_n || p.head_ldw % 4 != 0 ||
      !ga_aligned16(p.head_W) || p.c_cs != 1 || (p.c_rs & 3) != 0 ||
      !ga_aligned16(p.C) || (p.bias && !ga_aligned16(p.bias)) || p.H || p.accum ||
      p.colsum || p.epi != EPI_BIAS_ACT || p.K < 1)
    return 1;
  p.gx = p.M / bm; p.gy = 1; p.gz = 1;
  dim3 grid((unsigned)p.gx);
  // algorithmic flops of both layers
  const double flops = 2.0 * (double)p.M * (double)p.N * ((double)p.K + p.head_n);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ga_prof_events(GA_PROF_GEMM_NT_128, flops, &e0, &e1);
  if (p.N == 64)
    hipExtLaunchKernelGGL((gemm_f32_kernel<64, 64, 2, 2, true, true, BK, true>), grid,
                          dim3(256), 0, stream, e0, e1, 0, p);
  else if (p.N == 128)
    hipExtLaunchKernelGGL((gemm_f32_kernel<64, 128, 1, 4, true, true, BK, true>), grid,
                          dim3(256), 0, stream, e0, e1, 0, p);
  else
    hipExtLaunchKernelGGL((gemm_f32_kernel<64, 256, 1, 8, true, true, BK, true>), grid,
                          dim3(512), 0, stream, e0, e1, 0, p);
  GA_CHECK_LAUNCH("gemm_f32 (+head)");
  return GA_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI (see include/garage_amd.h)
// ---------------------------------------------------------------------------
struct ga_mlp_desc {
  int32_t n_layers;    // linear layers (hidden + output), 1..8
  int32_t dims[9];     // dims[0] = input width, dims[l + 1] = output width of layer l
  int64_t w_off[8];    // offset (floats) of W_l [dims[l+1]][round4(dims[l])] in params
  int64_t b_off[8];    // offset (floats) of b_l [dims[l+1]]
  int64_t act_off[8];  // offset (floats) of layer l's output in the activation
                       // workspace, row stride round4(dims[l + 1]) (hidden layers)
  int32_t hidden_act;  // 0 tanh, 1 relu, 2 none
  int32_t output_act;  // 0 none, 1 tanh, 2 relu (forward codes)
  int32_t layer_norm;  // 1: LayerNorm in front of every hidden linear layer
  int32_t pad_;
  int64_t ln_off[8];   // gamma_l [round4(dims[l])] in params, beta_l right behind it
  int64_t lnx_off[8];  // normalised input of hidden layer l in the activation
                       // workspace, row stride round4(dims[l])
  int64_t lns_off[8];  // (mean, rstd) per row of hidden layer l's input, there too
};

// lnorm.hip
int ga_ln_forward(const float* X, int64_t ldx, const int32_t* idx, int64_t M, int D,
                  const float* gamma, const float* beta, float* Y, int64_t ldy,
                  float* stats, hipStream_t stream);
int ga_ln_backward(float* dY, int64_t ldd, const float* X, int64_t ldx, const int32_t* idx,
                   const float* stats, int64_t M, int D, const float* gamma, int want_dx,
                   int hact, int rows_per_split, int n_splits, float* dgamma, float* dbeta,
                   int64_t split_stride, hipStream_t stream);
int ga_ln_jvp(const float* tX, int64_t ldt, const float* X, int64_t ldx, const int32_t* idx,
              const float* stats, int64_t M, int D, const float* gamma,
              const float* tgamma, const float* tbeta, float* tY, int64_t ldy,
              hipStream_t stream);

// The whole-network forward in one launch (policy_fused.hip) for nets whose
// layers fit its LDS tiles; ga_set_fused_forward(0) forces the per-layer GEMMs.
extern "C" int ga_policy_step_fused_supported(const ga_mlp_desc* d);
extern "C" int ga_mlp_forward_fused_f32(const ga_mlp_desc* d, const float* params,
                                        const float* X, int64_t ldx,
                                        const int32_t* row_idx, int64_t M,
                                        float* acts, float* out, int64_t ldo,
                                        hipStream_t stream);
// Off by default: at the C3 minibatch (32768 x 256 x 256) the fused forward
// measures 86-107 us against 82-87 us for the three per-layer GEMMs -- it keeps
// one workgroup per CU (140 KB of LDS) and its per-layer epilogues are exposed,
// which costs what the saved activation round trip gains.  The rollout step
// (policy_step_fused_kernel, n_envs rows) is where the fusion pays.
// Outputs-only forward of a whole two-hidden-layer tanh network in one launch
// (fused_train.hip: mlp_eval_forward_kernel); ga_set_eval_forward(0) makes callers
// that ask fall back to the per-layer kernels.
static int g_eval_forward = -1;
extern "C" int ga_set_eval_forward(int on) {
  g_eval_forward = on != 0;
  return GA_OK;
}
extern "C" int ga_mlp_forward_eval_supported(const ga_mlp_desc* d) {
  if (g_eval_forward < 0) {
    const char* e = getenv("GARAGE_AMD_EVAL_FORWARD");
    g_eval_forward = e ? atoi(e) != 0 : 1;
  }
  // (64-wide layers: the per-layer kernels' 64 x 64 tiles are 2 % faster at C2)
  return g_eval_forward && d && d->n_layers == 3 && d->hidden_act == 0 &&
         d->output_act == 0 && !d->layer_norm && d->dims[1] >= 128 && d->dims[2] >= 128 &&
         ga_fused_eval_supported(3, d->dims);
}
static int g_fused_forward = 0;
extern "C" int ga_set_fused_forward(int on) {
  g_fused_forward = on != 0;
  return 0;
}

// Streaming kernels for the layer products with one dimension <= 32 (skinny.hip).
// Return 1 when they do not take the shape: the MFMA tile kernel handles it.
int ga_skinny_forward(const float* X, int64_t ldx, const int32_t* idx, const float* W,
                      int64_t ldw, bool w_kc, const float* bias, int act,
                      const float* H, int64_t ldh, float* Y, int64_t ldy, int M, int N,
                      int K, hipStream_t stream);
int ga_skinny_wgrad(const float* Wd, int64_t ldw, const int32_t* w_idx, const float* Nr,
                    int64_t ldn, const int32_t* n_idx, int rows, int wide, int NS,
                    int rows_per_split, int n_splits, float* C, int64_t c_wide_stride,
                    int64_t c_narrow_stride, int64_t split_stride, float* colsum_wide,
                    float* colsum_narrow, const float* Wn, int64_t ldwn, float* dz_out,
                    int64_t lddz, hipStream_t stream);
// 0 off, 1 hidden layers up to 128 wide, 2 also 256-wide ones.  At 256 units the
// fused launch (64 x 256 tiles, 75 KB of LDS) saves 7.7 us per minibatch with the
// chip to itself (C3, one stream: 167.0 -> 160.6 ms per iteration) but loses 1 %
// when the policy and value chains share the chip on two streams, where the narrow
// head GEMM it replaces was hidden under the other chain's kernels anyway
// (3 x A/B: 147.1 / 148.5 / 149.4 vs 146.6 / 146.7 / 147.5 ms) -- so the default
// stops at 128 and both schedules keep the same arithmetic.
static int g_fuse_head_forward = 1;
extern "C" int ga_set_fused_head_forward(int mode) {
  g_fuse_head_forward = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return 0;
}
extern "C" int ga_set_small_m_gemm(int on) {
  g_small_m = on != 0;
  return 0;
}
static int g_fuse_head_dgrad = 1;
extern "C" int ga_set_fused_head_dgrad(int on) {
  g_fuse_head_dgrad = on != 0;
  return 0;
}
static int g_skinny = 1;
extern "C" int ga_set_skinny_kernels(int on) {
  g_skinny = on != 0;
  return 0;
}

static int check_desc(const ga_mlp_desc* d, const char* who) {
  GA_REQUIRE(d != nullptr, "%s: null descriptor", who);
  GA_REQUIRE(d->n_layers >= 1 && d->n_layers <= 8, "%s: n_layers %d not in 1..8",
             who, d->n_layers);
  for (int l = 0; l <= d->n_layers; ++l)
    GA_REQUIRE(d->dims[l] >= 1, "%s: dims[%d] < 1", who, l);
  for (int l = 0; l < d->n_layers; ++l)
    GA_REQUIRE(d->w_off[l] % 4 == 0 && d->act_off[l] % 4 == 0,
               "%s: offsets of layer %d not 16-B aligned", who, l);
  GA_REQUIRE(d->hidden_act >= 0 && d->hidden_act <= 6, "%s: hidden_act %d not in 0..6",
             who, d->hidden_act);
  GA_REQUIRE(d->output_act >= 0 && d->output_act <= 6, "%s: output_act %d not in 0..6",
             who, d->output_act);
  if (d->layer_norm)
    for (int l = 0; l + 1 < d->n_layers; ++l)
      GA_REQUIRE(d->ln_off[l] % 4 == 0 && d->lnx_off[l] % 4 == 0 && d->dims[l] <= 1024,
                 "%s: layer normalisation of layer %d: unaligned offsets or more than "
                 "1024 inputs", who, l);
  return GA_OK;
}

extern "C" int ga_mlp_forward_f32(const ga_mlp_desc* d, const float* params,
                                  const float* X, int64_t ldx,
                                  const int32_t* row_idx, int64_t M, float* acts,
                                  float* out, int64_t ldo, hipStream_t stream) {
  int rc = check_desc(d, "ga_mlp_forward_f32");
  if (rc) return rc;
  GA_REQUIRE(params && X, "ga_mlp_forward_f32: null pointer");
  // out == NULL: hidden layers only (the head is fused into the loss kernel)
  GA_REQUIRE(out || d->n_layers >= 2, "ga_mlp_forward_f32: nothing to compute");
  // acts == NULL: outputs only (ga_mlp_forward_eval_supported: the whole network in
  // one launch, no activation reaches memory)
  if (!acts && d->n_layers > 1) {
    GA_REQUIRE(out && ga_mlp_forward_eval_supported(d),
               "ga_mlp_forward_f32: acts workspace needed");
    GA_REQUIRE(M >= 0 && M < (1ll << 31), "ga_mlp_forward_f32: bad M");
    if (M == 0) return GA_OK;
    return ga_fused_eval_forward(X, ldx, row_idx, M, d->dims, params + d->w_off[0],
                                 params + d->b_off[0], params + d->w_off[1],
                                 params + d->b_off[1], params + d->w_off[2],
                                 params + d->b_off[2], out, ldo, stream);
  }
  GA_REQUIRE(M >= 0 && M < (1ll << 31), "ga_mlp_forward_f32: bad M");
  GA_REQUIRE(ldx % 4 == 0 && ldx >= d->dims[0], "ga_mlp_forward_f32: ldx %lld",
             (long long)ldx);
  GA_REQUIRE(!out || ldo >= d->dims[d->n_layers], "ga_mlp_forward_f32: ldo too small");
  GA_REQUIRE(ga_aligned16(params) && ga_aligned16(X) && (!acts || ga_aligned16(acts)),
             "ga_mlp_forward_f32: pointers must be 16-B aligned");
  if (M == 0) return GA_OK;
  if (out && g_fused_forward && d->hidden_act == 0 && d->output_act == 0 &&
      !d->layer_norm && ga_policy_step_fused_supported(d))
    return ga_mlp_forward_fused_f32(d, params, X, ldx, row_idx, M, acts, out, ldo,
                                    stream);
  const int L = d->n_layers;
  for (int l = 0; l < L; ++l) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    if (l == 0) {
      p.A = X; p.lda = ldx; p.a_idx = row_idx;
    } else {
      p.A = acts + d->act_off[l - 1]; p.lda = round4(d->dims[l]);
    }
    p.B = params + d->w_off[l];
    p.ldb = round4(d->dims[l]);
    const bool last = (l == L - 1);
    if (last && !out) break;
    if (d->layer_norm && !last) {
      // LayerNorm(prev) -> Linear -> nonlinearity
      // (multi_headed_mlp_module.py:77-92): the GEMM reads the normalised rows
      const int64_t ldn = round4(d->dims[l]);
      float* xn = acts + d->lnx_off[l];
      rc = ga_ln_forward(p.A, p.lda, p.a_idx, M, d->dims[l], params + d->ln_off[l],
                         params + d->ln_off[l] + ldn, xn, ldn, acts + d->lns_off[l],
                         stream);
      if (rc) return rc;
      p.A = xn; p.lda = ldn; p.a_idx = nullptr;
    }
    p.C = last ? out : acts + d->act_off[l];
    p.c_rs = last ? ldo : round4(d->dims[l + 1]);
    p.c_cs = 1;
    p.M = (int)M; p.N = d->dims[l + 1]; p.K = d->dims[l];
    p.epi = EPI_BIAS_ACT;
    p.bias = params + d->b_off[l];
    p.act = last ? d->output_act : act_forward_code(d->hidden_act);
    p.k_per_split = (int)ga_ceil_div(p.K, BK) * BK;
    // (the streaming kernels know tanh and the identity)
    if (g_skinny && p.act <= 1 && p.K <= 32 && p.N > 32) {
      rc = ga_skinny_forward(p.A, p.lda, p.a_idx, p.B, p.ldb, true, p.bias, p.act,
                             nullptr, 0, p.C, p.c_rs, p.M, p.N, p.K, stream);
      if (rc < 0) return rc;
      if (rc == 0) continue;
    }
    if (out && l == L - 2 && d->output_act == 0 && !d->layer_norm &&
        (g_fuse_head_forward == 2 || (g_fuse_head_forward == 1 && p.N <= 128))) {
      p.head_W = params + d->w_off[L - 1];
      p.head_ldw = round4(d->dims[L - 1]);
      p.head_bias = params + d->b_off[L - 1];
      p.head_n = d->dims[L];
      p.head_out = out;
      p.head_ld = ldo;
      rc = launch_gemm_with_head(p, stream);
      if (rc < 0) return rc;
      if (rc == 0) break;  // both layers done
      p.head_n = 0;
    }
    rc = launch_gemm<true, true>(p, 1, stream);
    if (rc) return rc;
  }
  return GA_OK;
}

extern "C" int64_t ga_mlp_backward_splits(const ga_mlp_desc* d, int64_t M) {
  // Rows of the batch each weight-gradient workgroup reduces before writing a
  // slab: large enough to amortise the slab write, small enough to fill 256 CUs.
  // 256 rows per slab: the widest layer (256x256 -> 2x2 tiles) then launches
  // 4 * M/256 workgroups, i.e. 512 at the C3 minibatch of 32768 rows.
  int64_t s = ga_ceil_div(M, 256);
  // nets whose layers are all <= 64 wide have one weight-gradient tile per split:
  // 128-row slabs double the workgroups (their slabs are a few KB each)
  bool small = true;
  for (int l = 0; l <= d->n_layers; ++l) small = small && d->dims[l] <= 64;
  if (small) s = ga_ceil_div(M, 128);
  // wide layers have many output tiles per split: fewer, longer splits then fill the
  // chip just as well, and every split less is a slab of the whole parameter vector
  // not written and not read back (C5, 512-wide layers: 16 tiles per split; 128 splits
  // of 512 rows moved 744 MB of slabs per optimizer step, 64 splits of 1024 rows --
  // 1024 workgroups for the widest layer -- move half).  256 x 256 layers (4 tiles)
  // keep 128 splits.
  int64_t tiles = 1;
  for (int l = 0; l < d->n_layers; ++l) {
    const int64_t t = ga_ceil_div(d->dims[l + 1], 128) * ga_ceil_div(d->dims[l], 128);
    tiles = t > tiles ? t : tiles;
  }
  static int64_t target_env = -1;  // workgroups of the widest layer's weight gradient
  if (target_env < 0) {
    const char* e = getenv("GARAGE_AMD_WGRAD_WORKGROUPS");  // developer sweep
    target_env = e ? atoll(e) : 0;
    if (target_env < 1) target_env = 0;
  }
  // (the split-operand weight-gradient kernel is three times faster per row: half the
  // workgroups and half the slabs -- 64 splits at C3 -- are the better trade there,
  // measured 82.3 -> 79.8 ms per iteration; an engine keeps the split count it was
  // built with)
  const int64_t target = target_env ? target_env : (ga_split_bf16_enabled() ? 256 : 1024);
  int64_t by_tiles = ga_ceil_div(target, tiles);
  // (never below 64 splits on that account: the streaming weight-gradient kernels of the
  // narrow layers take one workgroup per split and column block)
  if (!target_env && ga_split_bf16_enabled() && by_tiles < 64) by_tiles = 64;
  if (!small && s > by_tiles) s = by_tiles;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  return s;
}

extern "C" int ga_mlp_backward_range_f32(const ga_mlp_desc* d, const float* params,
                                         const float* X, int64_t ldx,
                                         const int32_t* row_idx, int64_t M,
                                         const float* acts, const float* dout,
                                         int64_t ldo, float* dacts, float* grad_slabs,
                                         int64_t slab_stride, int64_t n_splits,
                                         int l_start, int fused_first,
                                         hipStream_t stream);

// dW = dz^T in (+ db = column sums of dz) of the MIDDLE layer of two 3-layer networks,
// both out_w x in_w with 33 .. wide sides (the 128 x 128-tile kernel), split-K over the
// M rows into n_splits slabs each: the launch ga_mlp_backward_range_f32 makes for
// layer 1 with fused_first = 1, for two networks in one grid.  Same tiles, same k
// ranges, same summation order per element.
extern "C" int ga_wgrad_mid_pair(int64_t M, int64_t n_splits, int out_w, int in_w,
                                 const float* dza, const float* ina, float* slabs_wa,
                                 float* slabs_ba, int64_t slab_stride_a,
                                 const float* dzb, const float* inb, float* slabs_wb,
                                 float* slabs_bb, int64_t slab_stride_b,
                                 hipStream_t stream) {
  GA_REQUIRE(dza && ina && slabs_wa && slabs_ba && dzb && inb && slabs_wb && slabs_bb,
             "ga_wgrad_mid_pair: null pointer");
  GA_REQUIRE(M > 0 && M < (1ll << 31) && n_splits >= 1 && n_splits <= 1024 &&
                 out_w > 64 && in_w > 64 && slab_stride_a % 4 == 0 &&
                 slab_stride_b % 4 == 0,
             "ga_wgrad_mid_pair: unsupported shape");
  GA_REQUIRE(ga_aligned16(dza) && ga_aligned16(ina) && ga_aligned16(slabs_wa) &&
                 ga_aligned16(dzb) && ga_aligned16(inb) && ga_aligned16(slabs_wb),
             "ga_wgrad_mid_pair: pointers must be 16-B aligned");
  const int kps = (int)(ga_ceil_div(ga_ceil_div(M, n_splits), BK) * BK);
  GemmPair pp;
  const float* dz[2] = {dza, dzb};
  const float* in[2] = {ina, inb};
  float* sw[2] = {slabs_wa, slabs_wb};
  float* sb[2] = {slabs_ba, slabs_bb};
  const int64_t ss[2] = {slab_stride_a, slab_stride_b};
  for (int i = 0; i < 2; ++i) {
    GemmParams& p = i ? pp.b : pp.a;
    memset(&p, 0, sizeof(p));
    p.K = (int)M;
    p.k_per_split = kps;
    p.epi = EPI_PLAIN;
    p.c_split_stride = ss[i];
    p.colsum = sb[i];
    p.colsum_split_stride = ss[i];
    p.A = dz[i]; p.lda = round4(out_w); p.B = in[i]; p.ldb = round4(in_w);
    p.M = out_w; p.N = in_w;
    p.C = sw[i]; p.c_rs = round4(in_w); p.c_cs = 1;
    p.colsum_of_b = 0;
    p.gx = (int)ga_ceil_div(p.M, 128); p.gy = (int)ga_ceil_div(p.N, 128);
    p.gz = (int)n_splits;
  }
  const double flops = 2.0 * 2.0 * (double)out_w * (double)in_w * (double)M;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ga_prof_events(GA_PROF_GEMM_TN_128, flops, &e0, &e1);
  ga_prof_count(GA_PROF_GEMM_TN_128);
  const dim3 grid((unsigned)(2 * pp.a.gx * pp.a.gy * pp.a.gz));
  hipExtLaunchKernelGGL((gemm_f32_pair_kernel<128, 128, 2, 4, false, false>), grid,
                        dim3(512), 0, stream, e0, e1, 0, pp);
  GA_CHECK_LAUNCH("gemm_f32_pair");
  return GA_OK;
}

extern "C" int ga_mlp_backward_f32(const ga_mlp_desc* d, const float* params,
                                   const float* X, int64_t ldx,
                                   const int32_t* row_idx, int64_t M,
                                   const float* acts, const float* dout,
                                   int64_t ldo, float* dacts, float* grad_slabs,
                                   int64_t slab_stride, int64_t n_splits,
                                   hipStream_t stream) {
  GA_REQUIRE(d != nullptr, "ga_mlp_backward_f32: null descriptor");
  return ga_mlp_backward_range_f32(d, params, X, ldx, row_idx, M, acts, dout, ldo, dacts,
                                   grad_slabs, slab_stride, n_splits, d->n_layers - 1, 0,
                                   stream);
}

// Layers l_start .. 0 (fused_train.h).  l_start = n_layers - 1 with `dout` is the
// whole backward pass; the fused optimizer step enters below the head with the
// data gradient of the last hidden layer already in `dacts`.
extern "C" int ga_mlp_backward_range_f32(const ga_mlp_desc* d, const float* params,
                                         const float* X, int64_t ldx,
                                         const int32_t* row_idx, int64_t M,
                                         const float* acts, const float* dout,
                                         int64_t ldo, float* dacts, float* grad_slabs,
                                         int64_t slab_stride, int64_t n_splits,
                                         int l_start, int fused_first,
                                         hipStream_t stream) {
  int rc = check_desc(d, "ga_mlp_backward_f32");
  if (rc) return rc;
  GA_REQUIRE(params && X && grad_slabs, "ga_mlp_backward_f32: null pointer");
  GA_REQUIRE(l_start >= 0 && l_start < d->n_layers &&
                 (l_start < d->n_layers - 1 || dout) && (!fused_first || l_start >= 1),
             "ga_mlp_backward_f32: bad layer range");
  GA_REQUIRE(d->n_layers == 1 || (acts && dacts),
             "ga_mlp_backward_f32: workspaces needed");
  GA_REQUIRE(M > 0 && M < (1ll << 31), "ga_mlp_backward_f32: bad M");
  GA_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && slab_stride % 4 == 0,
             "ga_mlp_backward_f32: strides must be multiples of 4");
  GA_REQUIRE(n_splits >= 1 && n_splits <= 1024, "ga_mlp_backward_f32: n_splits");
  GA_REQUIRE(ga_aligned16(params) && ga_aligned16(X) && (!dout || ga_aligned16(dout)) &&
                 ga_aligned16(grad_slabs) && (!acts || ga_aligned16(acts)) &&
                 (!dacts || ga_aligned16(dacts)),
             "ga_mlp_backward_f32: pointers must be 16-B aligned");
  const int L = d->n_layers;
  int kps = (int)(ga_ceil_div(ga_ceil_div(M, n_splits), BK) * BK);
  for (int l = l_start; l >= (fused_first ? 1 : 0); --l) {
    bool dgrad_done = fused_first && l == 1;
    const float* dz = (l == L - 1) ? dout : dacts + d->act_off[l];
    const int64_t lddz = (l == L - 1) ? ldo : round4(d->dims[l + 1]);
    const int out_w = d->dims[l + 1], in_w = d->dims[l];
    // ---- weight + bias gradient slabs: dW[o][i] = sum_b dz[b][o] * in[b][i]
    {
      GemmParams p;
      memset(&p, 0, sizeof(p));
      const bool ln = d->layer_norm && l < L - 1;  // this layer reads normalised rows
      const float* in = ln ? acts + d->lnx_off[l]
                           : ((l == 0) ? X : acts + d->act_off[l - 1]);
      const int64_t ldin = (ln || l > 0) ? round4(in_w) : ldx;
      const int32_t* in_idx = (l == 0 && !ln) ? row_idx : nullptr;
      p.K = (int)M;
      p.k_per_split = kps;
      p.epi = EPI_PLAIN;
      p.c_split_stride = slab_stride;
      p.colsum = grad_slabs + d->b_off[l];
      p.colsum_split_stride = slab_stride;
      if (in_w <= 32 && out_w > 32) {
        // (out x in), narrow in: natural orientation, 256x32 tiles
        p.A = dz; p.lda = lddz; p.B = in; p.ldb = ldin; p.b_idx = in_idx;
        p.M = out_w; p.N = in_w;
        p.C = grad_slabs + d->w_off[l]; p.c_rs = round4(in_w); p.c_cs = 1;
        p.colsum_of_b = 0;
      } else if (out_w <= 32) {
        // narrow out: compute dW^T = in^T dz so the narrow side is N
        p.A = in; p.lda = ldin; p.a_idx = in_idx; p.B = dz; p.ldb = lddz;
        p.M = in_w; p.N = out_w;
        p.C = grad_slabs + d->w_off[l]; p.c_rs = 1; p.c_cs = round4(in_w);
        p.colsum_of_b = 1;
      } else {
        p.A = dz; p.lda = lddz; p.B = in; p.ldb = ldin; p.b_idx = in_idx;
        p.M = out_w; p.N = in_w;
        p.C = grad_slabs + d->w_off[l]; p.c_rs = round4(in_w); p.c_cs = 1;
        p.colsum_of_b = 0;
      }
      rc = 1;
      if (g_skinny && in_w <= 32 && out_w > 32) {
        // wide = dz (bias gradient = its column sums), narrow = layer input
        rc = ga_skinny_wgrad(dz, lddz, nullptr, in, ldin, in_idx, (int)M, out_w, in_w, kps,
                             (int)n_splits, grad_slabs + d->w_off[l], round4(in_w), 1,
                             slab_stride, grad_slabs + d->b_off[l], nullptr, nullptr, 0,
                             nullptr, 0, stream);
      } else if (g_skinny && out_w <= 32 && in_w > 32) {
        // head layer: the same pass over the hidden activations also yields the
        // data gradient of the layer below (it needs dz and tanh' of `in` only)
        const bool with_dz = g_fuse_head_dgrad && l > 0 && in_idx == nullptr &&
                             d->hidden_act == 0 && !d->layer_norm;
        rc = ga_skinny_wgrad(in, ldin, in_idx, dz, lddz, nullptr, (int)M, in_w, out_w, kps,
                             (int)n_splits, grad_slabs + d->w_off[l], 1, round4(in_w),
                             slab_stride, nullptr, grad_slabs + d->b_off[l],
                             with_dz ? params + d->w_off[l] : nullptr, round4(in_w),
                             with_dz ? dacts + d->act_off[l - 1] : nullptr, round4(in_w),
                             stream);
        if (rc == 1 && with_dz)  // shape not taken with the data gradient: without
          rc = ga_skinny_wgrad(in, ldin, in_idx, dz, lddz, nullptr, (int)M, in_w, out_w,
                               kps, (int)n_splits, grad_slabs + d->w_off[l], 1,
                               round4(in_w), slab_stride, nullptr,
                               grad_slabs + d->b_off[l], nullptr, 0, nullptr, 0, stream);
        else if (rc == 0 && with_dz)
          dgrad_done = true;
      }
      if (rc < 0) return rc;
      if (rc == 1) {
        rc = launch_gemm<false, false>(p, (int)n_splits, stream);
        if (rc) return rc;
      }
    }
    // ---- data gradient for the layer below
    // A normalised layer input (hidden layers with layer_norm) takes the plain
    // product dz W -- also for the first layer, whose gamma / beta need it -- and
    // the LayerNorm's backward pass then turns it, in place, into the data
    // gradient of the layer below.
    const bool ln_in = d->layer_norm && l < L - 1;
    if ((l > 0 || ln_in) && !dgrad_done) {
      GemmParams p;
      memset(&p, 0, sizeof(p));
      float* dst = l > 0 ? dacts + d->act_off[l - 1] : dacts + d->lnx_off[0];
      p.A = dz; p.lda = lddz;
      p.B = params + d->w_off[l]; p.ldb = round4(in_w);
      p.C = dst; p.c_rs = round4(in_w); p.c_cs = 1;
      p.M = (int)M; p.N = in_w; p.K = out_w;
      if (ln_in) {
        p.epi = EPI_PLAIN;
      } else {
        p.epi = EPI_MUL_DTANH;
        p.H = acts + d->act_off[l - 1]; p.ldh = round4(in_w);
        p.hact = d->hidden_act;
      }
      p.k_per_split = (int)ga_ceil_div(p.K, BK) * BK;
      rc = 1;
      if (g_skinny && !ln_in && d->hidden_act == 0 && p.K <= 32 && p.N > 32)
        rc = ga_skinny_forward(p.A, p.lda, nullptr, p.B, p.ldb, false, nullptr, 0, p.H,
                               p.ldh, p.C, p.c_rs, p.M, p.N, p.K, stream);
      if (rc < 0) return rc;
      if (rc == 1) {
        rc = launch_gemm<true, false>(p, 1, stream);
        if (rc) return rc;
      }
      if (ln_in) {
        const int64_t ldn = round4(in_w);
        rc = ga_ln_backward(dst, ldn, l == 0 ? X : acts + d->act_off[l - 1],
                            l == 0 ? ldx : ldn, l == 0 ? row_idx : nullptr,
                            acts + d->lns_off[l], M, in_w, params + d->ln_off[l],
                            l > 0 ? 1 : 0, d->hidden_act, kps, (int)n_splits,
                            grad_slabs + d->ln_off[l], grad_slabs + d->ln_off[l] + ldn,
                            slab_stride, stream);
        if (rc) return rc;
      }
    }
  }
  return GA_OK;
}

// Tangent (forward-mode) pass: with dtheta = `tangent` (flat parameter layout) and
// the activations of a forward at the same rows in `acts`,
//   tz_l = in_l dW_l^T + db_l + tin_l W_l^T,   th_l = tz_l * (1 - h_l^2)
// (in_0 = X, tin_0 = 0); `tout` receives d(output).  This is the J v half of the
// Fisher-vector product the TRPO policy step solves with
// (torch/optimizers/conjugate_gradient_optimizer.py:18-66 takes the same product
// by double backward through the KL constraint).
extern "C" int ga_mlp_jvp_f32(const ga_mlp_desc* d, const float* params,
                              const float* tangent, const float* X, int64_t ldx,
                              const int32_t* row_idx, int64_t M, const float* acts,
                              float* tacts, float* tout, int64_t ldo,
                              hipStream_t stream) {
  int rc = check_desc(d, "ga_mlp_jvp_f32");
  if (rc) return rc;
  GA_REQUIRE(params && tangent && X && tout, "ga_mlp_jvp_f32: null pointer");
  GA_REQUIRE(d->n_layers == 1 || (acts && tacts), "ga_mlp_jvp_f32: workspaces needed");
  GA_REQUIRE(M > 0 && M < (1ll << 31), "ga_mlp_jvp_f32: bad M");
  GA_REQUIRE(ldx % 4 == 0 && ldx >= d->dims[0] && ldo >= d->dims[d->n_layers],
             "ga_mlp_jvp_f32: leading dimensions");
  GA_REQUIRE(ga_aligned16(params) && ga_aligned16(tangent) && ga_aligned16(X) &&
                 (!acts || ga_aligned16(acts)) && (!tacts || ga_aligned16(tacts)),
             "ga_mlp_jvp_f32: pointers must be 16-B aligned");
  const int L = d->n_layers;
  for (int l = 0; l < L; ++l) {
    const bool last = (l == L - 1);
    const int in_w = d->dims[l], out_w = d->dims[l + 1];
    float* C = last ? tout : tacts + d->act_off[l];
    const int64_t ldc = last ? ldo : round4(out_w);
    const float* H = last ? nullptr : acts + d->act_off[l];
    // a normalised layer input: its tangent (through the LayerNorm, from the
    // tangent of the layer below and of gamma / beta) is a second product even
    // for the first layer
    const bool ln = d->layer_norm && !last;
    const int64_t ldn = round4(in_w);
    if (ln) {
      rc = ga_ln_jvp(l > 0 ? tacts + d->act_off[l - 1] : nullptr, ldn,
                     l > 0 ? acts + d->act_off[l - 1] : X, l > 0 ? ldn : ldx,
                     l > 0 ? nullptr : row_idx, acts + d->lns_off[l], M, in_w,
                     params + d->ln_off[l], tangent + d->ln_off[l],
                     tangent + d->ln_off[l] + ldn, tacts + d->lnx_off[l], ldn, stream);
      if (rc) return rc;
    }
    const bool two = l > 0 || ln;
    // in_l dW_l^T + db_l  (and the tanh' factor when it is the only product)
    GemmParams p;
    memset(&p, 0, sizeof(p));
    if (ln) {
      p.A = acts + d->lnx_off[l]; p.lda = ldn;
    } else if (l == 0) {
      p.A = X; p.lda = ldx; p.a_idx = row_idx;
    } else {
      p.A = acts + d->act_off[l - 1]; p.lda = round4(in_w);
    }
    p.B = tangent + d->w_off[l]; p.ldb = round4(in_w);
    p.C = C; p.c_rs = ldc; p.c_cs = 1;
    p.M = (int)M; p.N = out_w; p.K = in_w;
    p.epi = EPI_BIAS_ACT; p.bias = tangent + d->b_off[l]; p.act = 0;
    if (!two) { p.H = H; p.ldh = ldc; p.hact = d->hidden_act; }
    p.k_per_split = (int)ga_ceil_div(p.K, BK) * BK;
    rc = launch_gemm<true, true>(p, 1, stream);
    if (rc) return rc;
    if (two) {
      // += tin_l W_l^T, then the tanh' factor
      GemmParams q;
      memset(&q, 0, sizeof(q));
      q.A = ln ? tacts + d->lnx_off[l] : tacts + d->act_off[l - 1];
      q.lda = round4(in_w);
      q.B = params + d->w_off[l]; q.ldb = round4(in_w);
      q.C = C; q.c_rs = ldc; q.c_cs = 1;
      q.M = (int)M; q.N = out_w; q.K = in_w;
      q.accum = 1;
      if (H) { q.epi = EPI_MUL_DTANH; q.H = H; q.ldh = ldc; q.hact = d->hidden_act; }
      else q.epi = EPI_PLAIN;
      q.k_per_split = (int)ga_ceil_div(q.K, BK) * BK;
      rc = launch_gemm<true, true>(q, 1, stream);
      if (rc) return rc;
    }
  }
  return GA_OK;
}

namespace {
__global__ __launch_bounds__(256) void act_slope_mul_kernel(float* dout, int64_t ldd,
                                                            const float* out, int64_t ldo,
                                                            int64_t M, int N, int act) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * N) return;
  const int64_t i = e / N;
  const int j = (int)(e % N);
  const float o = out[i * ldo + j];
  dout[i * ldd + j] *= act_slope_fwd(o, act);  // (forward codes, gemm_core.h)
}
}  // namespace

extern "C" int ga_act_slope_mul_f32(float* dout, int64_t ldd, const float* out,
                                    int64_t ldo, int64_t M, int N, int act,
                                    hipStream_t stream) {
  GA_REQUIRE(dout && out && M >= 0 && N >= 1 && ldd >= N && ldo >= N && act >= 0 &&
                 act <= 6,
             "ga_act_slope_mul_f32: bad arguments");
  if (M == 0 || act == 0) return GA_OK;
  hipLaunchKernelGGL(act_slope_mul_kernel, dim3((unsigned)ga_ceil_div(M * N, 256)),
                     dim3(256), 0, stream, dout, ldd, out, ldo, M, N, act);
  GA_CHECK_LAUNCH("act_slope_mul");
  return GA_OK;
}

// Plain GEMM entry used by tests: C[M,N] = A[M,K] * B[N,K]^T (both k-contiguous).
extern "C" int ga_gemm_nt_f32(const float* A, int64_t lda, const float* B,
                              int64_t ldb, float* C, int64_t ldc, int64_t M,
                              int64_t N, int64_t K, hipStream_t stream) {
  GA_REQUIRE(A && B && C, "ga_gemm_nt_f32: null pointer");
  GA_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ga_aligned16(A) && ga_aligned16(B),
             "ga_gemm_nt_f32: operands must be 16-B aligned with ld %% 4 == 0");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.c_rs = ldc; p.c_cs = 1;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.epi = EPI_PLAIN;
  p.k_per_split = (int)ga_ceil_div(K, BK) * BK;
  return launch_gemm<true, true>(p, 1, stream);
}
