// One optimizer step's forward + loss + backward of a NARROW network in one launch
// (gfx950): two tanh hidden layers of equal width H = 32 or 64, <= 32 inputs,
// <= 8 outputs -- the CartPole-class networks of BASELINE.json configs[0-1]
// (MLP(32,32), MLP(64,64)) at minibatches of any size.
//
// At these widths every weight of the network fits in LDS many times over (18 KB at
// H = 64), so a workgroup takes 64 rows of the minibatch through the WHOLE step --
// gather, both hidden layers, head, loss row by row with its gradient seed
// (loss_rows.h: torch/algos/ppo.py:96-132, vpg.py:434-454,
// gaussian_mlp_value_function.py:81-98), both data gradients, all three weight
// gradients -- with the activations of its rows never leaving the CU, and writes
// only its share of the gradient (a few KB).  reduce_regions_adam_kernel
// (fused_train.hip) then sums the shares in a fixed order, finishes the loss and
// applies Adam: two launches per optimizer step of VPG._train_policy /
// _train_value_function (vpg.py:250-293) instead of five (fused kernels) or ten
// (per-layer kernels), each of which was latency bound at these shapes.
//
// All products with K >= 32 run on v_mfma_f32_32x32x2_f32 from LDS operands (one
// 32x32 tile per wave at H = 64); sums over the 64 rows of a tile are taken in a
// fixed order, so results are bitwise reproducible; they agree with the other
// paths to rounding.
#include "common.h"
#include <hip/hip_ext.h>

#include "prof.h"

#include "fused_train.h"
#include "loss_rows.h"

// ---- Audit (round 3) of out-of-range lanes / idle waves: no clamped "dead" loads in
// this kernel.  Every global load is behind its own bounds test -- sample row
// m = min(m0 + lane, M - 1) (wave 0 only); observation quads under m0 + rr < M &&
// 4 q < ld0; W1 / W2 / W_head staged by loops bounded by the layer's own sizes
// (n < H, 4 q < ld0 resp. q < H / 4, j < A); biases under tid < H resp. tid < A.  The
// waves that have no tile at H = 32 (tile_on == false) touch LDS only.  Partial sums
// go to part + blockIdx.x * stride with stride = ga_narrow_step_stride (checked
// against exactly-sized buffers by tests/host/update_loop_harness.cpp, check 9).
namespace {

constexpr int NS_ROWS = 64;
constexpr int NS_THREADS = 256;
constexpr int NS_HN = 8;

typedef float ns_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float ns_tanh(float x) {
  return ga_tanh(x);  // common.h
}

struct NarrowParams {
  const float* params;
  int64_t w_off[3], b_off[3];
  int in_w, out_w, M;
  const float* X; int64_t ldx;
  LossRowArgs loss;
  float* part;      // [tiles][stride]: dW1 [H][ld0], db1 [H], dW2 [H][H], db2 [H],
  int64_t stride;   //                  dW_head [8][H], db_head [8]
  double* lpart;    // [tiles][2]
  long long* dbg;   // developer hook: phase timestamps of workgroup 0
};

#define NS_STAMP(i) \
  if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) p.dbg[i] = wall_clock64()

// One 32 x 32 output tile on the matrix cores: acc += sum_k A(i, k) B(k, j), K a
// multiple of 8.  Lane l feeds A(l % 32, k) and B(k, l % 32) with k = 8 g + 4 (l / 32)
// + q in MFMA q of group g (any k <-> slot map is valid as long as A and B agree).
//   A_KC: A(i, k) = A[i * lda + k] (one 16-B read per group)   else A[k * lda + i]
//   B_KC: B(k, j) = B[j * ldb + k]                            else B[k * ldb + j]
template <bool A_KC, bool B_KC>
__device__ __forceinline__ ns_f32x16 ns_tile(const float* A, int lda, const float* B,
                                             int ldb, int K, int lane) {
  const int l31 = lane & 31, half = lane >> 5;
  ns_f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (int g = 0; g < K / 8; ++g) {
    float a[4], b[4];
    const int k0 = 8 * g + 4 * half;
    if (A_KC) {
      const float4 v = *reinterpret_cast<const float4*>(A + l31 * lda + k0);
      a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = A[(k0 + q) * lda + l31];
    }
    if (B_KC) {
      const float4 v = *reinterpret_cast<const float4*>(B + l31 * ldb + k0);
      b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = B[(k0 + q) * ldb + l31];
    }
    // two accumulators alternate: consecutive MFMAs do not wait for each other
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] += acc1[i];
  return acc0;
}

// NS_LDX = floats per staged observation / first-layer weight row: round8(in_w) + 4
// (12, 20 or 36).  At 12 and 20 the workgroup stays under 80 KB of LDS, so a
// workgroup of the policy pass and one of the value-function pass (two streams)
// share a CU.
template <int H, int NS_LDX>
// (2 waves per SIMD where the LDS tiles allow two workgroups per CU: one of each
// network's chain -- without the bound the compiler takes 254 + 64 registers and
// the two chains' launches can only take turns)
__global__ __launch_bounds__(NS_THREADS, NS_LDX <= 20 ? 2 : 1) void narrow_train_kernel(
    NarrowParams p) {
  constexpr int LDH = H + 4;
  constexpr int CT = H / 32;        // column tiles of a hidden layer
  constexpr int QW = H / 4;         // hidden columns per wave in the VALU phases
  // dz2 is written in P5; until then its memory holds the first-layer weights (read
  // in P1 only) and the four planes of the head's partial sums (P3)
  constexpr int DZ2_FLOATS = NS_ROWS * LDH;
  constexpr int W1_FLOATS = H * NS_LDX, PLANES_FLOATS = 4 * NS_ROWS * NS_HN;
  constexpr int SCRATCH = DZ2_FLOATS > W1_FLOATS + PLANES_FLOATS
                              ? DZ2_FLOATS : W1_FLOATS + PLANES_FLOATS;
  __shared__ __attribute__((aligned(16))) float xs[NS_ROWS * NS_LDX];
  __shared__ __attribute__((aligned(16))) float w2s[H * LDH];
  __shared__ __attribute__((aligned(16))) float whs[NS_HN * H];
  __shared__ __attribute__((aligned(16))) float h1[NS_ROWS * LDH];
  __shared__ __attribute__((aligned(16))) float h2[NS_ROWS * LDH];   // later dZ1
  __shared__ __attribute__((aligned(16))) float scratch[SCRATCH];
  __shared__ __attribute__((aligned(16))) float outl[NS_ROWS * NS_HN];
  __shared__ __attribute__((aligned(16))) float doutl[NS_ROWS * NS_HN];
  __shared__ float b1s[H], b2s[H], bhs[NS_HN];
  float* dz2 = scratch;
  float* w1s = scratch;
  float* planes = scratch + W1_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * NS_ROWS;
  const int M = p.M, in_w = p.in_w, A = p.out_w;
  const int ld0 = (in_w + 3) & ~3;
  const int K1 = (in_w + 7) & ~7;  // first-layer reduction length, zero padded
  const LossRowArgs& L = p.loss;
  const float* W1 = p.params + p.w_off[0];
  const float* W2 = p.params + p.w_off[1];
  const float* Wh = p.params + p.w_off[2];

  NS_STAMP(0);
  // ---- the sample of this lane's row (wave 0 computes the loss rows)
  float act[8];
  float adv = 0.f, old_ll = 0.f, ret = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) act[j] = 0.f;
  const bool live = m0 + lane < M;
  if (wave == 0) {
    const int m = min(m0 + lane, M - 1);
    const int64_t src = L.idx ? (int64_t)L.idx[m] : (int64_t)m;
    if (L.kind == 1) {
      ret = L.returns[src];
    } else {
      adv = L.adv[src];
      if (L.algo != 1) old_ll = L.old_ll[src];
      const float* arow = L.actions + src * L.lda;
      if (L.kind == 2) {
        act[0] = arow[0];
      } else {
        const float4 a0 = *reinterpret_cast<const float4*>(arow);
        act[0] = a0.x; act[1] = a0.y; act[2] = a0.z; act[3] = a0.w;
        if (A > 4) {
          const float4 a1 = *reinterpret_cast<const float4*>(arow + 4);
          act[4] = a1.x; act[5] = a1.y; act[6] = a1.z; act[7] = a1.w;
        }
      }
    }
  }
  // ---- P0: stage the rows' observations and every weight (zero padded)
  for (int e = tid; e < NS_ROWS * (NS_LDX / 4); e += NS_THREADS) {
    const int rr = e / (NS_LDX / 4), q = e % (NS_LDX / 4);
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + rr < M && 4 * q < ld0) {
      const int m = m0 + rr;
      const int64_t src = L.idx ? (int64_t)L.idx[m] : (int64_t)m;
      x = *reinterpret_cast<const float4*>(p.X + src * p.ldx + 4 * q);
      x.x = (4 * q + 0 < in_w) ? x.x : 0.f;
      x.y = (4 * q + 1 < in_w) ? x.y : 0.f;
      x.z = (4 * q + 2 < in_w) ? x.z : 0.f;
      x.w = (4 * q + 3 < in_w) ? x.w : 0.f;
    }
    *reinterpret_cast<float4*>(xs + rr * NS_LDX + 4 * q) = x;
  }
  for (int e = tid; e < H * (NS_LDX / 4); e += NS_THREADS) {
    const int n = e / (NS_LDX / 4), q = e % (NS_LDX / 4);
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * q < ld0) w = *reinterpret_cast<const float4*>(W1 + (int64_t)n * ld0 + 4 * q);
    *reinterpret_cast<float4*>(w1s + n * NS_LDX + 4 * q) = w;
  }
  for (int e = tid; e < H * (H / 4); e += NS_THREADS) {
    const int n = e / (H / 4), q = e % (H / 4);
    *reinterpret_cast<float4*>(w2s + n * LDH + 4 * q) =
        *reinterpret_cast<const float4*>(W2 + (int64_t)n * H + 4 * q);
  }
  for (int e = tid; e < NS_HN * (H / 4); e += NS_THREADS) {
    const int j = e / (H / 4), q = e % (H / 4);
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < A) w = *reinterpret_cast<const float4*>(Wh + (int64_t)j * H + 4 * q);
    *reinterpret_cast<float4*>(whs + j * H + 4 * q) = w;
  }
  if (tid < H) {
    b1s[tid] = p.params[p.b_off[0] + tid];
    b2s[tid] = p.params[p.b_off[1] + tid];
  }
  if (tid < NS_HN) bhs[tid] = tid < A ? p.params[p.b_off[2] + tid] : 0.f;
  __syncthreads();
  NS_STAMP(1);

  // tile of this wave in the 2 x CT tilings below (H = 32: waves 2, 3 idle there)
  const int tri = wave & 1, tcj = wave >> 1;
  const bool tile_on = tcj < CT;

  // ---- P1: H1 = tanh(X W1^T + b1)
  if (tile_on) {
    const ns_f32x16 acc = ns_tile<true, true>(xs + 32 * tri * NS_LDX, NS_LDX,
                                              w1s + 32 * tcj * NS_LDX, NS_LDX, K1, lane);
    const int col = 32 * tcj + l31;
    const float b = b1s[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * tri + (r & 3) + 8 * (r >> 2) + 4 * half;
      h1[row * LDH + col] = ns_tanh(acc[r] + b);
    }
  }
  __syncthreads();
  NS_STAMP(2);
  // ---- P2: H2 = tanh(H1 W2^T + b2)
  if (tile_on) {
    const ns_f32x16 acc = ns_tile<true, true>(h1 + 32 * tri * LDH, LDH,
                                              w2s + 32 * tcj * LDH, LDH, H, lane);
    const int col = 32 * tcj + l31;
    const float b = b2s[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * tri + (r & 3) + 8 * (r >> 2) + 4 * half;
      h2[row * LDH + col] = ns_tanh(acc[r] + b);
    }
  }
  __syncthreads();
  NS_STAMP(3);
  // ---- P3: head outputs (lane = row, wave = a quarter of the hidden columns)
  {
    float a[NS_HN];
#pragma unroll
    for (int j = 0; j < NS_HN; ++j) a[j] = 0.f;
#pragma unroll
    for (int i = 0; i < QW / 4; ++i) {
      const float4 h = *reinterpret_cast<const float4*>(h2 + lane * LDH + wave * QW + 4 * i);
#pragma unroll
      for (int j = 0; j < NS_HN; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(whs + j * H + wave * QW + 4 * i);
        a[j] = fmaf(h.x, w.x, a[j]); a[j] = fmaf(h.y, w.y, a[j]);
        a[j] = fmaf(h.z, w.z, a[j]); a[j] = fmaf(h.w, w.w, a[j]);
      }
    }
    float* mine = planes + (wave * NS_ROWS + lane) * NS_HN;
    *reinterpret_cast<float4*>(mine) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(mine + 4) = make_float4(a[4], a[5], a[6], a[7]);
  }
  __syncthreads();
  for (int o = tid; o < NS_ROWS * NS_HN; o += NS_THREADS) {
    float s = bhs[o % NS_HN];
#pragma unroll
    for (int w = 0; w < 4; ++w) s += planes[w * NS_ROWS * NS_HN + o];
    outl[o] = s;
  }
  __syncthreads();
  NS_STAMP(4);
  // ---- P4: loss rows (wave 0)
  if (wave == 0) {
    float s = 0.f, inv_var = 1.f;
    if (L.kind != 2) {
      s = *L.log_std;
      if (L.kind == 0)
        s = ga_log_std(s, L.has_min, L.min_log_std, L.has_max, L.max_log_std, nullptr);
      inv_var = expf(-2.f * s);
    }
    float out[8], dout[8];
    const float4 o0 = *reinterpret_cast<const float4*>(outl + lane * NS_HN);
    const float4 o1 = *reinterpret_cast<const float4*>(outl + lane * NS_HN + 4);
    out[0] = o0.x; out[1] = o0.y; out[2] = o0.z; out[3] = o0.w;
    out[4] = o1.x; out[5] = o1.y; out[6] = o1.z; out[7] = o1.w;
    double second = 0.0;
    double first = lr_row(L, s, inv_var, out, act, adv, old_ll, ret, dout, &second);
    if (!live) {
      first = 0.0; second = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) dout[j] = 0.f;
    }
    *reinterpret_cast<float4*>(doutl + lane * NS_HN) =
        make_float4(dout[0], dout[1], dout[2], dout[3]);
    *reinterpret_cast<float4*>(doutl + lane * NS_HN + 4) =
        make_float4(dout[4], dout[5], dout[6], dout[7]);
    first = ga_wave_sum(first);
    second = ga_wave_sum(second);
    if (lane == 0) {
      p.lpart[2 * blockIdx.x + 0] = first;
      p.lpart[2 * blockIdx.x + 1] = second;
    }
  }
  __syncthreads();
  NS_STAMP(5);
  float* part = p.part + (int64_t)blockIdx.x * p.stride;
  float* pW1 = part;
  float* pb1 = pW1 + (int64_t)H * ld0;
  float* pW2 = pb1 + H;
  float* pb2 = pW2 + (int64_t)H * H;
  float* pWh = pb2 + H;
  float* pbh = pWh + NS_HN * H;
  // ---- P5: dZ2 = (dout W_head) (1 - H2^2)   (lane = row, wave = column quarter)
  {
    const float4 d0 = *reinterpret_cast<const float4*>(doutl + lane * NS_HN);
    const float4 d1 = *reinterpret_cast<const float4*>(doutl + lane * NS_HN + 4);
    const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
    for (int i = 0; i < QW / 4; ++i) {
      const int c = wave * QW + 4 * i;
      const float4 h = *reinterpret_cast<const float4*>(h2 + lane * LDH + c);
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < NS_HN; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(whs + j * H + c);
        z.x = fmaf(dd[j], w.x, z.x); z.y = fmaf(dd[j], w.y, z.y);
        z.z = fmaf(dd[j], w.z, z.z); z.w = fmaf(dd[j], w.w, z.w);
      }
      z.x *= (1.f - h.x * h.x); z.y *= (1.f - h.y * h.y);
      z.z *= (1.f - h.z * h.z); z.w *= (1.f - h.w * h.w);
      *reinterpret_cast<float4*>(dz2 + lane * LDH + c) = z;
    }
  }
  // ---- P6: head weight / bias gradient shares
  {
    constexpr int GROUPS = NS_THREADS / H, JPG = NS_HN / GROUPS;
    const int c = tid % H, j0 = (tid / H) * JPG;
    float g[JPG];
#pragma unroll
    for (int jj = 0; jj < JPG; ++jj) g[jj] = 0.f;
    for (int r = 0; r < NS_ROWS; ++r) {
      const float h = h2[r * LDH + c];
#pragma unroll
      for (int jj = 0; jj < JPG; ++jj) g[jj] = fmaf(doutl[r * NS_HN + j0 + jj], h, g[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < JPG; ++jj) pWh[(j0 + jj) * H + c] = g[jj];
    if (tid < NS_HN) {
      float b = 0.f;
      for (int r = 0; r < NS_ROWS; ++r) b += doutl[r * NS_HN + tid];
      pbh[tid] = b;
    }
  }
  __syncthreads();  // dz2 complete; every read of h2 done
  NS_STAMP(6);
  // ---- P7: dW2[n][k] = sum_r dZ2[r][n] H1[r][k]   (CT x CT tiles)
  for (int t = wave; t < CT * CT; t += 4) {
    const int tn = t / CT, tk = t % CT;
    const ns_f32x16 acc = ns_tile<false, false>(dz2 + 32 * tn, LDH, h1 + 32 * tk, LDH,
                                                NS_ROWS, lane);
    const int k = 32 * tk + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * tn + (r & 3) + 8 * (r >> 2) + 4 * half;
      pW2[n * H + k] = acc[r];
    }
  }
  // ---- P8: dZ1 = (dZ2 W2) (1 - H1^2) -> the h2 buffer
  if (tile_on) {
    const ns_f32x16 acc = ns_tile<true, false>(dz2 + 32 * tri * LDH, LDH, w2s + 32 * tcj,
                                               LDH, H, lane);
    const int col = 32 * tcj + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * tri + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float h = h1[row * LDH + col];
      h2[row * LDH + col] = acc[r] * (1.f - h * h);
    }
  }
  // bias gradient of layer 2 (column sums of dZ2)
  if (tid >= NS_THREADS - H) {
    const int n = tid - (NS_THREADS - H);
    float b = 0.f;
    for (int r = 0; r < NS_ROWS; ++r) b += dz2[r * LDH + n];
    pb2[n] = b;
  }
  __syncthreads();
  NS_STAMP(7);
  // ---- P9: dW1[n][k] = sum_r dZ1[r][n] X[r][k], db1
  {
    constexpr int GROUPS = NS_THREADS / H;
    const int n = tid % H, g0 = tid / H;
    for (int q = g0; 4 * q < ld0; q += GROUPS) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < NS_ROWS; ++r) {
        const float d = h2[r * LDH + n];
        const float4 x = *reinterpret_cast<const float4*>(xs + r * NS_LDX + 4 * q);
        g.x = fmaf(d, x.x, g.x); g.y = fmaf(d, x.y, g.y);
        g.z = fmaf(d, x.z, g.z); g.w = fmaf(d, x.w, g.w);
      }
      *reinterpret_cast<float4*>(pW1 + (int64_t)n * ld0 + 4 * q) = g;
    }
    if (tid >= NS_THREADS - H) {
      const int nn = tid - (NS_THREADS - H);
      float b = 0.f;
      for (int r = 0; r < NS_ROWS; ++r) b += h2[r * LDH + nn];
      pb1[nn] = b;
    }
  }
  __syncthreads();
  NS_STAMP(8);
}

}  // namespace

static long long* g_ns_dbg = nullptr;
// developer hook: phase timestamps (100 MHz wall clock) of workgroup 0 of the most
// recent launch -- first call arms it, second call reads 16 values back
extern "C" int ga_narrow_step_debug(long long* host_out16) {
  if (!g_ns_dbg) {
    if (hipMalloc(&g_ns_dbg, 16 * sizeof(long long)) != hipSuccess) return -1;
    (void)hipMemset(g_ns_dbg, 0, 16 * sizeof(long long));
    return 1;
  }
  (void)hipDeviceSynchronize();
  return hipMemcpy(host_out16, g_ns_dbg, 16 * sizeof(long long), hipMemcpyDeviceToHost) ==
                 hipSuccess ? 0 : -1;
}

// LossRowArgs from the epoch loop's arguments (fused_train.hip holds the same
// conversion for its kernels)
static LossRowArgs narrow_loss_args(const ga_fused_loss_args* l, int64_t M) {
  LossRowArgs L;
  memset(&L, 0, sizeof(L));
  L.kind = l->kind; L.actions = l->actions; L.lda = l->lda; L.old_ll = l->old_ll;
  L.adv = l->adv; L.returns = l->returns; L.idx = l->idx; L.log_std = l->log_std;
  L.has_min = l->has_min; L.has_max = l->has_max; L.min_log_std = l->min_log_std;
  L.max_log_std = l->max_log_std; L.A = l->A; L.algo = l->algo; L.clip = l->clip;
  L.ent_coeff = l->ent_coeff; L.ent_regularized = l->ent_flags & 1;
  L.ent_softplus = (l->ent_flags >> 1) & 1; L.ent_stop_grad = (l->ent_flags >> 2) & 1;
  L.double_softmax = l->double_softmax;
  L.invM = 1.f / (float)M;
  return L;
}

extern "C" int ga_narrow_step_supported(int n_layers, const int* dims) {
  return n_layers == 3 && dims[1] == dims[2] && (dims[1] == 32 || dims[1] == 64) &&
         dims[0] >= 1 && dims[0] <= 32 && dims[3] >= 1 && dims[3] <= 8;
}

extern "C" int64_t ga_narrow_step_stride(int in_w, int H) {
  const int64_t ld0 = (in_w + 3) & ~3;
  return (int64_t)H * ld0 + H + (int64_t)H * H + H + 8 * (int64_t)H + 8;
}

extern "C" int ga_narrow_train_step(const float* params, const int64_t* w_off,
                                    const int64_t* b_off, int in_w, int H, int out_w,
                                    const float* X, int64_t ldx, int64_t M,
                                    const ga_fused_loss_args* loss, float* part,
                                    double* lpart, hipStream_t stream) {
  GA_REQUIRE(params && w_off && b_off && X && loss && part && lpart,
             "ga_narrow_train_step: null pointer");
  const int dims[4] = {in_w, H, H, out_w};
  GA_REQUIRE(ga_narrow_step_supported(3, dims) && M >= 1 && M < (1ll << 31),
             "ga_narrow_train_step: unsupported shape");
  GA_REQUIRE(ga_aligned16(params) && ga_aligned16(X) && ga_aligned16(part) &&
                 ldx % 4 == 0 && ldx >= ((in_w + 3) & ~3) && w_off[0] % 4 == 0 &&
                 w_off[1] % 4 == 0 && w_off[2] % 4 == 0,
             "ga_narrow_train_step: operands must be 16-B aligned quads");
  GA_REQUIRE(loss->kind == 1 ? loss->returns != nullptr
                             : (loss->actions && loss->adv &&
                                (loss->algo == 1 || loss->old_ll) &&
                                (loss->kind == 2 ||
                                 (loss->lda % 4 == 0 && ga_aligned16(loss->actions) &&
                                  loss->lda >= ((loss->A + 3) & ~3)))),
             "ga_narrow_train_step: missing / misaligned minibatch arrays");
  GA_REQUIRE(loss->kind == 2 || loss->log_std, "ga_narrow_train_step: log_std");
  GA_REQUIRE(loss->algo == 0 || loss->algo == 1, "ga_narrow_train_step: algo");
  NarrowParams p;
  memset(&p, 0, sizeof(p));
  p.params = params;
  for (int l = 0; l < 3; ++l) { p.w_off[l] = w_off[l]; p.b_off[l] = b_off[l]; }
  p.in_w = in_w; p.out_w = out_w; p.M = (int)M; p.X = X; p.ldx = ldx;
  p.loss = narrow_loss_args(loss, M);
  p.part = part; p.stride = ga_narrow_step_stride(in_w, H); p.lpart = lpart;
  p.dbg = g_ns_dbg;
  const dim3 grid((unsigned)ga_fused_tiles(M));
  // algorithmic flops: forward + both backward products of every layer
  const double flops =
      6.0 * (double)M * ((double)in_w * H + (double)H * H + (double)H * out_w);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ga_prof_events(GA_PROF_NARROW_STEP, flops, &e0, &e1);
#define GA_NARROW_LAUNCH(HH, LL)                                                      \
  hipExtLaunchKernelGGL((narrow_train_kernel<HH, LL>), grid, dim3(NS_THREADS), 0, stream, \
                        e0, e1, 0, p)
  const int ldx_s = in_w <= 8 ? 12 : (in_w <= 16 ? 20 : 36);
  if (H == 64) {
    if (ldx_s == 12) GA_NARROW_LAUNCH(64, 12);
    else if (ldx_s == 20) GA_NARROW_LAUNCH(64, 20);
    else GA_NARROW_LAUNCH(64, 36);
  } else {
    if (ldx_s == 12) GA_NARROW_LAUNCH(32, 12);
    else if (ldx_s == 20) GA_NARROW_LAUNCH(32, 20);
    else GA_NARROW_LAUNCH(32, 36);
  }
#undef GA_NARROW_LAUNCH
  GA_CHECK_LAUNCH("narrow_train");
  return GA_OK;
}
