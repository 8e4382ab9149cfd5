// One optimizer step of a small minibatch in ONE launch.
//
// The reference's default minibatch is 64 samples (torch/algos/ppo.py:65-76), so
// VPG._train_policy / _train_value_function (vpg.py:250-293) take thousands of
// optimizer steps per iteration, each a chain of ten dependent launches of 5-10 us
// on the per-layer path.  Here the whole step -- gather, forward, loss, backward,
// Adam -- runs in one grid of H / 16 workgroups: workgroup g owns hidden columns
// [16 g, 16 g + 16) of both hidden layers (their weight rows, bias entries and
// the matching columns of the head), the 64 rows of the minibatch live in LDS, and
// the two places where a workgroup needs the other workgroups' columns (the
// second hidden activation before the head, its data gradient before the first
// layer's) are two grid barriers around 64 KB exchanges through L2:
//
//   A  X -> H1 (all columns, every workgroup: K <= 32)    -> own columns of H2
//      ---- barrier 1: H2 complete ----
//   B  head, loss, d(out) (every workgroup: <= 8 outputs)  -> own columns of dZ2,
//      own rows of dW2 (K = 64 rows), own columns of dW_head
//      ---- barrier 2: dZ2 complete ----
//   C  own columns of dZ1 (needs W2[:, own], staged before barrier 2), own rows of
//      dW1, Adam on everything the workgroup owns.
//
// Weights a wave reads are wave-uniform (a wave = 64 rows x one column group), so
// they come through the scalar cache; activations are read from LDS once.
// Shapes: two tanh hidden layers of equal width H (multiple of 32, <= 256), input
// width <= 32, <= 8 outputs, <= 64 rows; Gaussian or categorical PPO / VPG objective
// (with the entropy options), or the value function's Gaussian NLL.  Everything else takes the
// per-layer path.  Same formulas as losses.hip / gemm.hip; sums are taken in a
// different order, so results agree to rounding, not bit for bit.
#include "common.h"
#include "small_step.h"

// ---- Audit (round 3) of global loads whose lane / wave index can lie outside the
// layer -- the class of the round-2 fault (idle waves of a 32-wide net prefetching
// optimizer state past a small network's parameter buffer).  A load here is either
// guarded by its own bounds test or issued from an index CLAMPED into the buffer (its
// value is then discarded: a conditionally written register array would live in
// scratch memory).  Every site, with the bound that makes it safe:
//   p.idx[r]                live ? ... : 0            (rows >= M read nothing)
//   p.idx[rr], X rows       rr = min(.., M - 1); q = min(.., ld0 / 4 - 1)
//   b0[min(tid, H - 1)]; b1[c0 + tid] for tid < 16 (c0 + 15 < H); bh / Wh guarded by
//                           tid - 16 < A, tid / 16 < A
//   W2 columns (wc)         e = min(.., H * 4 - 1)          -> n < H, c0 + 4 q + 3 < H
//   W1 (w0q)                e = min(.., H * ld0 / 4 - 1)
//   W2 own rows (w2q)       e = min(.., 16 * (H / 4) - 1)   -> row c0 + 15 < H
//   dZ2 exchange (dq)       e = min(.., 64 * (H / 4) - 1)   -> p.xdz[64][H]
//   head shares (pv)        g < gridDim.x ? ... : 0         -> p.xh2[H / 16][64][8]
//   Adam prefetch (ap/am/av) cj = min(cg + 4 t, H / 32 - 1)  (the round-2 fix)
// Largest flat index over all of them: w_off[1] + H * H - 1 and b_off[2] + A - 1, both
// inside the flat layout; the exchanges need 32 H and 64 H floats of the activation
// workspaces (update.cpp takes this path from 32 workspace rows up: 64 H floats).
// tests/host/update_loop_harness.cpp (check 9) restates these formulas against
// exactly-sized buffers under AddressSanitizer.
namespace {

constexpr int SS_ROWS = 64;
constexpr int SS_THREADS = 256;
constexpr int SS_COLS = 16;     // hidden columns per workgroup
constexpr int SS_HMAX = 256;
constexpr int SS_LDH = SS_HMAX + 4;
constexpr int SS_LDX = 36;
constexpr int SS_LDO = 20;      // own-slice tiles [64][16 + 4]
constexpr double SS_HALF_LOG_2PI = 0.91893853320467274178;

typedef const __attribute__((address_space(4))) float* uptr;  // wave-uniform reads

struct SmallStepParams {
  // network (flat parameter layout of engine.py)
  float* params; float* m; float* v;
  int64_t w_off[3], b_off[3];
  int in_w, H, out_w, M;
  // minibatch
  const float* X; int64_t ldx; const int32_t* idx;
  int kind;  // 0 Gaussian policy, 1 value function, 2 categorical policy
  int double_softmax;
  const float* actions; int64_t lda; const float* old_ll; const float* adv;
  const float* returns;
  int algo; float clip;
  int has_min, has_max; float min_log_std, max_log_std;
  float ent_coeff; int ent_regularized, ent_softplus, ent_stop_grad;
  // Adam (one step for every parameter)
  float lerp_w, beta2, one_minus_beta2, neg_step_size, bc2_sqrt, eps;
  int learn_std;
  // exchange buffers [64][H] each (H2, dZ2) and the barrier words {count, flag}
  // (zero between launches)
  float* xh2; float* xdz; unsigned* bar;
  float* loss_out;
  int* fault;  // set when a barrier gave up (the launch then wrote no parameter)
  int max_polls;
  long long* dbg;  // optional: cycle counter of workgroup 0 at the phase boundaries
};

__device__ __forceinline__ float ss_tanh(float x) {
  return ga_tanh(x);  // common.h
}

// torch.optim.Adam, one element (losses.hip: adam_update)
__device__ __forceinline__ void ss_adam_math(const SmallStepParams& a, float g, float& p,
                                             float& m, float& v) {
#pragma clang fp contract(off)
  const float diff = g - m;
  m = fmaf(a.lerp_w, diff, m);
  const float gg = (a.one_minus_beta2 * g) * g;
  v = v * a.beta2 + gg;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  const float num = a.neg_step_size * m;
  p = p + num / denom;
}
__device__ __forceinline__ void ss_adam(const SmallStepParams& a, float g, int64_t i) {
  float p = a.params[i], m = a.m[i], v = a.v[i];
  ss_adam_math(a, g, p, m, v);
  a.params[i] = p; a.m[i] = m; a.v[i] = v;
}
// N elements `stride` apart: every load first (one memory round trip, not N)
template <int N>
__device__ __forceinline__ void ss_adam_n(const SmallStepParams& a, const float (&g)[N],
                                          int64_t i0, int64_t stride) {
  float p[N], m[N], v[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    p[n] = a.params[i0 + n * stride];
    m[n] = a.m[i0 + n * stride];
    v[n] = a.v[i0 + n * stride];
  }
#pragma unroll
  for (int n = 0; n < N; ++n) ss_adam_math(a, g[n], p[n], m[n], v[n]);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    a.params[i0 + n * stride] = p[n];
    a.m[i0 + n * stride] = m[n];
    a.v[i0 + n * stride] = v[n];
  }
}

// Grid barrier on {count, phase}: the last arrival clears the count and moves the
// phase word old_v -> new_v with a compare-and-swap; waiters poll until the word
// leaves old_v.  Two barriers per launch (0 -> 1, 1 -> 0) leave both words at 0.
// A waiter that runs out of polls (a grid that is not co-resident would never
// arrive) tries old_v -> SS_ABORT with the same compare-and-swap, so ONE atomic
// word decides for the whole grid whether a barrier was passed or the launch is
// abandoned: whoever swaps first wins and every workgroup reads the same verdict.
// An abandoned launch returns before any parameter or moment is written (all Adam
// writes sit behind barrier 2), raises *fault and leaves the phase word at
// SS_ABORT, which makes later launches return at their first barrier too: the
// parameters stay those of the last complete step until the host has looked
// (VPG._train_once checks the fault word and re-arms the barrier words).
constexpr unsigned SS_ABORT = 2u;
__device__ __forceinline__ bool ss_grid_barrier(const SmallStepParams& p, unsigned old_v,
                                                unsigned new_v, int* verdict) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned seen;
    const unsigned t = atomicAdd(p.bar, 1u);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(p.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      const unsigned prev = atomicCAS(p.bar + 1, old_v, new_v);
      seen = prev == old_v ? new_v : prev;
    } else {
      int polls = 0;
      while ((seen = __hip_atomic_load(p.bar + 1, __ATOMIC_ACQUIRE,
                                       __HIP_MEMORY_SCOPE_AGENT)) == old_v) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > p.max_polls) {
          const unsigned prev = atomicCAS(p.bar + 1, old_v, SS_ABORT);
          seen = prev == old_v ? SS_ABORT : prev;
          break;
        }
      }
    }
    if (seen != new_v) atomicExch(p.fault, 1);
    *verdict = seen == new_v;
  }
  __syncthreads();
  __threadfence();
  return *verdict != 0;
}

typedef float ss_f32x16 __attribute__((ext_vector_type(16)));

// out[64][16] = A[64][K] Bt[16][K]^T on the matrix cores: wave w takes row tile w & 1
// and the K half w >> 1; B is padded to a 32-column tile with zeros.  Both operands
// are k-contiguous in LDS (row strides lda, ldb: multiples of 4 floats), so a lane
// reads 4 consecutive k of its row with one 16-B load per 4 MFMAs (lane half h
// feeds k = 8 g + 4 h + q to step q of group g, as in gemm.hip); two accumulators
// alternate so that consecutive MFMAs do not wait for each other.  The two K halves
// land in red[2][64][16] (summed by the caller, half 0 first).
__device__ __forceinline__ void ss_mma_64xKx16(const float* A, int lda, const float* Bt,
                                               int ldb, int K, float* red) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
  const int w = threadIdx.x >> 6, ri = w & 1, half = w >> 1;
  ss_f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  const int k0 = half * (K / 2);
  const float* a = A + (32 * ri + l31) * lda + k0 + 4 * kh;
  const float* b = Bt + (l31 & 15) * ldb + k0 + 4 * kh;
  const bool b_on = l31 < 16;
  for (int g = 0; g < K / 16; ++g) {
    const float4 av = *reinterpret_cast<const float4*>(a + 8 * g);
    float4 bv = *reinterpret_cast<const float4*>(b + 8 * g);
    if (!b_on) bv = make_float4(0.f, 0.f, 0.f, 0.f);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc1, 0, 0, 0);
  }
  if (b_on) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32 * ri + (i & 3) + 8 * (i >> 2) + 4 * kh;
      red[(half * SS_ROWS + row) * SS_COLS + l31] = acc0[i] + acc1[i];
    }
  }
}

#define SS_MARK(i)                                                        \
  if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) p.dbg[i] = wall_clock64()

__global__ __launch_bounds__(SS_THREADS) void small_step_kernel(SmallStepParams p) {
  __shared__ __attribute__((aligned(16))) float full[SS_ROWS * SS_LDH];   // H1, later dZ2
  __shared__ __attribute__((aligned(16))) float xs[SS_ROWS * SS_LDX];
  __shared__ __attribute__((aligned(16))) float w1c[SS_COLS * SS_LDH];    // W2[:, own]^T
  __shared__ __attribute__((aligned(16))) float h1own[SS_ROWS * SS_LDO];
  __shared__ __attribute__((aligned(16))) float h2own[SS_ROWS * SS_LDO];
  __shared__ __attribute__((aligned(16))) float dzown[SS_ROWS * SS_LDO];  // dZ2, later dZ1
  __shared__ __attribute__((aligned(16))) float outl[SS_ROWS * 8];
  __shared__ __attribute__((aligned(16))) float doutl[SS_ROWS * 8];
  __shared__ __attribute__((aligned(16))) float part[4 * SS_ROWS * 8];
  __shared__ __attribute__((aligned(16))) float w0s[SS_HMAX * 32];        // W1 (all rows)
  __shared__ float bias_s[SS_HMAX + SS_COLS + 8];  // b1 (all), b2 (own), b_head
  __shared__ float whs[8 * SS_COLS];               // W_head[:, own] (0 beyond A)
  __shared__ float dlogstd_s;
  __shared__ int verdict_s;

  SS_MARK(0);
  const int tid = threadIdx.x;
  const int r = tid & 63;                                         // row = lane
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave
  const int H = p.H, in_w = p.in_w, A = p.out_w, M = p.M;
  const int ld0 = (in_w + 3) & ~3;
  const int c0 = blockIdx.x * SS_COLS;
  const float* W0 = p.params + p.w_off[0];
  const float* W1 = p.params + p.w_off[1];
  const float* Wh = p.params + p.w_off[2];
  const float* b0 = p.params + p.b_off[0];
  const float* b1 = p.params + p.b_off[1];
  const float* bh = p.params + p.b_off[2];
  const bool live = r < M;
  const int64_t src = live ? (p.idx ? (int64_t)p.idx[r] : (int64_t)r) : 0;

  // ---- A.0: the minibatch rows (rows >= M are zero) and W2's columns [c0, c0 + 16)
  {
    // 64 rows x 9 quads (36 floats, zero beyond in_w): at most 3 per thread; the
    // row ids first, then the quads (X rows are padded to ldx >= round4(in_w))
    int64_t srow[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int rr = min((tid + SS_THREADS * i) / (SS_LDX / 4), M - 1);
      srow[i] = p.idx ? (int64_t)p.idx[rr] : (int64_t)rr;
    }
    float4 xq[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int q = min((tid + SS_THREADS * i) % (SS_LDX / 4), ld0 / 4 - 1);
      xq[i] = *reinterpret_cast<const float4*>(p.X + srow[i] * p.ldx + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + SS_THREADS * i, rr = e / (SS_LDX / 4), q = e % (SS_LDX / 4);
      if (e < SS_ROWS * (SS_LDX / 4)) {
        float4 x = xq[i];
        const bool row_ok = rr < M;
        x.x = (row_ok && 4 * q + 0 < in_w) ? x.x : 0.f;
        x.y = (row_ok && 4 * q + 1 < in_w) ? x.y : 0.f;
        x.z = (row_ok && 4 * q + 2 < in_w) ? x.z : 0.f;
        x.w = (row_ok && 4 * q + 3 < in_w) ? x.w : 0.f;
        *reinterpret_cast<float4*>(xs + e * 4) = x;
      }
    }
  }
  {
    // every load first, then the LDS stores: one memory round trip, not one per
    // loop iteration (H <= 256: at most 4 + 8 quads per thread)
    float4 wc[4], w0q[8];
    // biases: thread t < H: b1[t]; H <= t' = t - ... (second loads below)
    const float bq0 = b0[min(tid, H - 1)];
    const float bq1 = tid < SS_COLS ? b1[c0 + tid] : (tid < SS_COLS + 8 && tid - SS_COLS < A
                                                          ? bh[tid - SS_COLS] : 0.f);
    const float whq = (tid < 8 * SS_COLS && tid / SS_COLS < A)
                          ? Wh[(int64_t)(tid / SS_COLS) * H + c0 + tid % SS_COLS]
                          : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (unconditional loads from clamped addresses: a conditionally written
      // register array would be kept in scratch memory)
      const int e = min(tid + SS_THREADS * i, H * (SS_COLS / 4) - 1);
      const int n = e / (SS_COLS / 4), q = e % (SS_COLS / 4);
      wc[i] = *reinterpret_cast<const float4*>(W1 + (int64_t)n * H + c0 + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = min(tid + SS_THREADS * i, H * ld0 / 4 - 1);
      w0q[i] = reinterpret_cast<const float4*>(W0)[e];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + SS_THREADS * i;
      if (e < H * (SS_COLS / 4)) {  // transposed: w1c[j][n] = W2[n][c0 + j]
        const int n = e / (SS_COLS / 4), j0 = 4 * (e % (SS_COLS / 4));
        w1c[(j0 + 0) * SS_LDH + n] = wc[i].x;
        w1c[(j0 + 1) * SS_LDH + n] = wc[i].y;
        w1c[(j0 + 2) * SS_LDH + n] = wc[i].z;
        w1c[(j0 + 3) * SS_LDH + n] = wc[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + SS_THREADS * i;
      if (e < H * ld0 / 4) reinterpret_cast<float4*>(w0s)[e] = w0q[i];
    }
    if (tid < H) bias_s[tid] = bq0;
    if (tid < SS_COLS + 8) bias_s[SS_HMAX + tid] = bq1;
    if (tid < 8 * SS_COLS) whs[tid] = whq;
  }
  __syncthreads();
  // W2's own rows [16][H] go to LDS (over the first-layer weights) once A.1 is done
  // with those; their loads are issued now
  float4 w2q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = min(tid + SS_THREADS * i, SS_COLS * (H / 4) - 1);
    w2q[i] = *reinterpret_cast<const float4*>(W1 + (int64_t)(c0 + e / (H / 4)) * H +
                                              4 * (e % (H / 4)));
  }
  SS_MARK(1);

  // ---- A.1: H1 = tanh(X W1^T + b1), all H columns, on the matrix cores
  //      (v_mfma_f32_32x32x2_f32: 2 row tiles x H / 32 column tiles, 4 per wave at
  //      H = 256; lane l feeds A[row l % 32][k l / 32] and B[k l / 32][col l % 32])
  {
    const int lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int n_tiles = 2 * (H / 32);
    for (int t = cg; t < n_tiles; t += 8) {
      const int t1 = t + 4;
      const bool two = t1 < n_tiles;
      const int ri0 = t & 1, cj0 = t >> 1, ri1 = t1 & 1, cj1 = two ? (t1 >> 1) : cj0;
      ss_f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      const float* a0 = xs + (32 * ri0 + l31) * SS_LDX + 4 * kh;
      const float* a1 = xs + (32 * ri1 + l31) * SS_LDX + 4 * kh;
      const float* w0p = w0s + (32 * cj0 + l31) * ld0 + 4 * kh;
      const float* w1p = w0s + (32 * cj1 + l31) * ld0 + 4 * kh;
      for (int g = 0; 8 * g < ld0; ++g) {
        // k = 8 g + 4 kh + q; weight rows are ld0 long: quads past the row end are
        // masked (xs is zero there too, but 0 x stale LDS could be NaN)
        const bool k_ok = 8 * g + 4 * kh < ld0;
        const float4 x0 = *reinterpret_cast<const float4*>(a0 + 8 * g);
        const float4 x1 = *reinterpret_cast<const float4*>(a1 + 8 * g);
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (k_ok) {
          v0 = *reinterpret_cast<const float4*>(w0p + 8 * g);
          v1 = *reinterpret_cast<const float4*>(w1p + 8 * g);
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, v0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, v1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, v0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, v1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.z, v0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.z, v1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.w, v0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.w, v1.w, acc1, 0, 0, 0);
      }
      {
        const int col = 32 * cj0 + l31;
        const float bias = bias_s[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = 32 * ri0 + (i & 3) + 8 * (i >> 2) + 4 * kh;
          full[row * SS_LDH + col] = ss_tanh(acc0[i] + bias);
        }
      }
      if (two) {
        const int col = 32 * cj1 + l31;
        const float bias = bias_s[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = 32 * ri1 + (i & 3) + 8 * (i >> 2) + 4 * kh;
          full[row * SS_LDH + col] = ss_tanh(acc1[i] + bias);
        }
      }
    }
  }
  __syncthreads();
  SS_MARK(2);

  // ---- A.2: own columns of H2 = tanh(H1 W2^T + b2) on the matrix cores
  float* w2r = w0s;  // [16][SS_LDH]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + SS_THREADS * i;
    if (e < SS_COLS * (H / 4))
      *reinterpret_cast<float4*>(w2r + (e / (H / 4)) * SS_LDH + 4 * (e % (H / 4))) = w2q[i];
  }
  __syncthreads();
  SS_MARK(11);
  // B(k, j) = W2[c0 + j][k] = w2r[j * SS_LDH + k]
  ss_mma_64xKx16(full, SS_LDH, w2r, SS_LDH, H, part);
  SS_MARK(12);
  __syncthreads();
  SS_MARK(13);
  {
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[j] = (part[r * SS_COLS + 4 * cg + j] +
                part[(SS_ROWS + r) * SS_COLS + 4 * cg + j]) +
               bias_s[SS_HMAX + 4 * cg + j];
    float4 o;
    o.x = ss_tanh(acc[0]); o.y = ss_tanh(acc[1]); o.z = ss_tanh(acc[2]);
    o.w = ss_tanh(acc[3]);
    *reinterpret_cast<float4*>(h2own + r * SS_LDO + 4 * cg) = o;
    // own columns of H1 are needed again after `full` is reused
    *reinterpret_cast<float4*>(h1own + r * SS_LDO + 4 * cg) =
        *reinterpret_cast<const float4*>(full + r * SS_LDH + c0 + 4 * cg);
    // this workgroup's share of the head: out_part[r][j] = sum over its 16 columns
    // (4 per wave, summed over the waves in wave order below)
    const float hv[4] = {o.x, o.y, o.z, o.w};
    __syncthreads();  // every thread has taken its sums out of `part`
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sacc = 0.f;
      if (j < A) {
        const float4 w = *reinterpret_cast<const float4*>(whs + j * SS_COLS + 4 * cg);
        sacc = fmaf(hv[0], w.x, sacc);
        sacc = fmaf(hv[1], w.y, sacc);
        sacc = fmaf(hv[2], w.z, sacc);
        sacc = fmaf(hv[3], w.w, sacc);
      }
      part[(cg * SS_ROWS + r) * 8 + j] = sacc;
    }
  }
  __syncthreads();
  for (int e = tid; e < SS_ROWS * 8; e += SS_THREADS) {
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) sacc += part[q * SS_ROWS * 8 + e];
    p.xh2[(int64_t)blockIdx.x * SS_ROWS * 8 + e] = sacc;
  }
  SS_MARK(3);
  if (!ss_grid_barrier(p, 0u, 1u, &verdict_s)) return;
  SS_MARK(4);

  // ---- B.1: head outputs (every workgroup): the workgroups' shares in workgroup
  //      order, plus the bias
  {
    float pv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g)
        pv[i][g] = g < (int)gridDim.x
                       ? p.xh2[(int64_t)g * SS_ROWS * 8 + tid + SS_THREADS * i]
                       : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + SS_THREADS * i, j = e & 7;
      float sacc = bias_s[SS_HMAX + SS_COLS + j];  // 0 beyond the outputs
#pragma unroll
      for (int g = 0; g < 16; ++g) sacc += pv[i][g];  // absent workgroups add 0
      outl[e] = sacc;
    }
  }
  __syncthreads();
  SS_MARK(5);

  // ---- B.2: loss and d(loss)/d(out), one lane per row (wave 0)
  if (tid < 64) {
    const float invM = 1.f / (float)M;
    float s = p.params[0], s_chain = 1.f;
    bool s_grad = true;
    double obj = 0.0, ds = 0.0;
    float dm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dm[j] = 0.f;
    if (p.kind == 0) {
      s = ga_log_std(s, p.has_min, p.min_log_std, p.has_max, p.max_log_std, &s_chain);
      s_grad = s_chain != 0.f;
      const float inv_var = expf(-2.f * s);
      const float lognorm = s + (float)SS_HALF_LOG_2PI;
      if (live) {
        const float* a = p.actions + src * p.lda;
        float ll = 0.f, q = 0.f;
        for (int j = 0; j < A; ++j) {
          const float d = a[j] - outl[r * 8 + j];
          const float z = d * d * inv_var;
          q += z;
          ll += -0.5f * z - lognorm;
        }
        const float adv = p.adv[src];
        float o, g;
        if (p.algo == 1) {
          o = ll * adv;
          g = adv;
        } else {
          const float ratio = expf(ll - p.old_ll[src]);
          const float lo = 1.f - p.clip, hi = 1.f + p.clip;
          const float rc = fminf(fmaxf(ratio, lo), hi);
          const float s1 = ratio * adv, s2 = rc * adv;
          o = fminf(s1, s2);
          const float g1 = adv * ratio;
          const float g2 = (ratio >= lo && ratio <= hi) ? adv * ratio : 0.f;
          g = (s1 < s2) ? g1 : ((s1 > s2) ? g2 : 0.5f * (g1 + g2));
        }
        obj = (double)o;
        const float scale = -g * invM * inv_var;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < A) dm[j] = scale * (a[j] - outl[r * 8 + j]);
        ds = (double)(-g * (q - (float)A));
      }
    } else if (p.kind == 2) {
      // categorical head (losses.hip: ppo_categorical_loss_kernel): the scores are
      // logits, or -- double_softmax -- their softmax is
      s_grad = false;  // no log-std parameter: its slot keeps a zero gradient
      if (live) {
        float sc[8], pr[8], lp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = outl[r * 8 + j];
        float mx = sc[0];
#pragma unroll
        for (int j = 1; j < 8; ++j)
          if (j < A) mx = fmaxf(mx, sc[j]);
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < A) den += expf(sc[j] - mx);
        float lse;
        if (!p.double_softmax) {
          lse = mx + logf(den);
        } else {
          float s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < A) s2 += expf(expf(sc[j] - mx) / den);
          lse = logf(s2);
        }
        const int a = (int)p.actions[src * p.lda];
        float ll = 0.f, Hent = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pr[j] = 0.f; lp[j] = 0.f;
          if (j < A) {
            pr[j] = expf(sc[j] - mx) / den;
            lp[j] = (p.double_softmax ? pr[j] : sc[j]) - lse;
            Hent -= expf(lp[j]) * lp[j];
            if (j == a) ll = lp[j];
          }
        }
        float Hs = Hent, dHs = 1.f;
        if (p.ent_softplus) {
          dHs = 1.f / (1.f + expf(-Hent));
          Hs = Hent > 20.f ? Hent : log1pf(expf(Hent));
        }
        const float adv = p.adv[src];
        float o, g;
        if (p.algo == 1) {
          o = ll * adv;
          g = adv;
        } else {
          const float ratio = expf(ll - p.old_ll[src]);
          const float lo = 1.f - p.clip, hi = 1.f + p.clip;
          const float rc = fminf(fmaxf(ratio, lo), hi);
          const float s1 = ratio * adv, s2 = rc * adv;
          o = fminf(s1, s2);
          const float g1 = adv * ratio;
          const float g2 = (ratio >= lo && ratio <= hi) ? adv * ratio : 0.f;
          g = (s1 < s2) ? g1 : ((s1 > s2) ? g2 : 0.5f * (g1 + g2));
        }
        if (p.ent_regularized) o += p.ent_coeff * Hs;
        obj = (double)o;
        const float cH = (p.ent_regularized && !p.ent_stop_grad) ? p.ent_coeff * dHs : 0.f;
        float dp[8];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          dp[j] = 0.f;
          if (j < A) {
            const float q = expf(lp[j]);
            dp[j] = g * ((j == a ? 1.f : 0.f) - q) - cH * q * (lp[j] + Hent);
            dot += dp[j] * pr[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < A)
            dm[j] = p.double_softmax ? -(pr[j] * (dp[j] - dot)) * invM : -dp[j] * invM;
      }
    } else {
      const float inv_var = expf(-2.f * s);
      if (live) {
        const float d = p.returns[src] - outl[r * 8];
        const float z = d * d * inv_var;
        obj = (double)(0.5f * z + s + (float)SS_HALF_LOG_2PI);
        ds = (double)(1.f - z);
        dm[0] = -d * inv_var * invM;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) doutl[r * 8 + j] = dm[j];
    obj = ga_wave_sum(obj);
    ds = ga_wave_sum(ds);
    if (tid == 0) {
      double mean = obj / (double)M;
      double dls = ds / (double)M;
      if (p.kind == 0 && p.ent_regularized) {
        // entropy of the Independent Normal, A (0.5 + 0.5 log 2 pi + s): the same for
        // every state (losses.hip: ppo_gaussian_finish)
        float ent = (float)A * (0.5f + (float)SS_HALF_LOG_2PI + s);
        float dent = (float)A;
        if (p.ent_softplus) {
          dent *= 1.f / (1.f + expf(-ent));
          ent = ent > 20.f ? ent : log1pf(expf(ent));
        }
        mean += (double)(p.ent_coeff * ent);
        if (!p.ent_stop_grad) dls += -(double)(p.ent_coeff * dent);
      }
      if (blockIdx.x == 0) *p.loss_out = (float)(p.kind == 1 ? mean : -mean);
      dlogstd_s = s_grad ? (float)dls * s_chain : 0.f;
    }
  }
  __syncthreads();
  SS_MARK(6);

  // ---- B.3: own columns of dZ2 = (d(out) W_head) (1 - H2^2); publish them
  {
    float dzp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o)  // rows >= A of whs are zero
        s = fmaf(doutl[r * 8 + o], whs[o * SS_COLS + 4 * cg + j], s);
      const float h = h2own[r * SS_LDO + 4 * cg + j];
      dzp[j] = s * (1.f - h * h);
    }
    const float4 dz = make_float4(dzp[0], dzp[1], dzp[2], dzp[3]);
    *reinterpret_cast<float4*>(dzown + r * SS_LDO + 4 * cg) = dz;
    *reinterpret_cast<float4*>(p.xdz + (int64_t)r * H + c0 + 4 * cg) = dz;
  }
  __syncthreads();

  // ---- B.4: gradients of what this workgroup owns of the head and of layer 2
  // (held in registers until every workgroup has read the old weights: phase C)
  float g_head = 0.f;       // threads < 8 * 16: dW_head[o][c0 + c]
  float g_bh = 0.f;         // workgroup 0, threads < 8
  if (tid < 8 * SS_COLS) {
    const int o = tid / SS_COLS, c = tid % SS_COLS;
    if (o < A)
      for (int rr = 0; rr < SS_ROWS; ++rr)
        g_head = fmaf(doutl[rr * 8 + o], h2own[rr * SS_LDO + c], g_head);
  }
  if (blockIdx.x == 0 && tid < 8 && tid < A)
    for (int rr = 0; rr < SS_ROWS; ++rr) g_bh += doutl[rr * 8 + tid];
  // dW2[c0 + n][k] = sum_r dZ2[r][c0 + n] H1[r][k] on the matrix cores: A(n, r) =
  // dzown[r][n] (rows n >= 16 of the tile are zero), B(r, k) = H1; wave w takes the
  // column tiles w, w + 4.  A lane ends up with rows {0..3, 8..11} + 4 (lane / 32)
  // of column 32 tile + lane % 32: g_w1[t][i], i < 8.
  float g_w1[2][8];
  {
    const int lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    ss_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const bool on0 = 32 * cg < H, on1 = 32 * (cg + 4) < H;
    const float* a = dzown + kh * SS_LDO + (l31 & 15);
    const float* b0p = full + kh * SS_LDH + 32 * cg + l31;
    const float* b1p = full + kh * SS_LDH + (on1 ? 32 * (cg + 4) : 32 * cg) + l31;
    const bool a_on = l31 < 16;
    if (on0) {
#pragma unroll 4
      for (int kk = 0; kk < SS_ROWS; kk += 2) {
        const float av = a_on ? a[kk * SS_LDO] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0p[kk * SS_LDH], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1p[kk * SS_LDH], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      g_w1[0][i] = acc0[i];
      g_w1[1][i] = on1 ? acc1[i] : 0.f;
    }
  }
  float g_b1 = 0.f;         // threads < 16: db2[c0 + tid]
  if (tid < SS_COLS)
    for (int rr = 0; rr < SS_ROWS; ++rr) g_b1 += dzown[rr * SS_LDO + tid];
  SS_MARK(7);
  if (!ss_grid_barrier(p, 1u, 0u, &verdict_s)) return;
  SS_MARK(8);

  // ---- C.1: all columns of dZ2 into LDS (over H1), then own columns of
  //      dZ1 = (dZ2 W2[:, own]) (1 - H1[:, own]^2)
  // 64 x H / 4 quads over 256 threads: at most 16 each, all in flight at once
  {
    float4 dq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = min(tid + SS_THREADS * i, SS_ROWS * (H / 4) - 1);
      dq[i] = *reinterpret_cast<const float4*>(p.xdz + (int64_t)(e / (H / 4)) * H +
                                               4 * (e % (H / 4)));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + SS_THREADS * i;
      if (e < SS_ROWS * (H / 4))
        *reinterpret_cast<float4*>(full + (e / (H / 4)) * SS_LDH + 4 * (e % (H / 4))) =
            dq[i];
    }
  }
  __syncthreads();
  SS_MARK(14);
  // the optimizer state of this workgroup's layer-2 rows: loads issued now, used
  // after the dZ1 product (C.2)
  float ap[2][8], am[2][8], av[2][8];
  {
    const int lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // waves whose column tile lies beyond the layer (H < 32 * (cg + 4 t) + 32) load
      // the layer's last tile instead: the values are not used, but the address
      // must stay inside the parameter buffer (a wave of a 32-wide net would
      // otherwise read up to 96 floats past its row -- past the end of a small
      // value network's buffer for the last rows)
      const int cj = min(cg + 4 * t, H / 32 - 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
        const int64_t e = p.w_off[1] + (int64_t)(c0 + row) * H + 32 * cj + l31;
        ap[t][i] = p.params[e]; am[t][i] = p.m[e]; av[t][i] = p.v[e];
      }
    }
  }
  // B(n, j) = W2[n][c0 + j] = w1c[j][n]
  ss_mma_64xKx16(full, SS_LDH, w1c, SS_LDH, H, part);
  __syncthreads();
  SS_MARK(15);
  {
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[j] = part[r * SS_COLS + 4 * cg + j] + part[(SS_ROWS + r) * SS_COLS + 4 * cg + j];
    float op[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float h = h1own[r * SS_LDO + 4 * cg + j];
      op[j] = acc[j] * (1.f - h * h);
    }
    const float4 o = make_float4(op[0], op[1], op[2], op[3]);
    __syncthreads();  // dzown (dZ2) was read by B.4 of this workgroup only: reuse it
    *reinterpret_cast<float4*>(dzown + r * SS_LDO + 4 * cg) = o;
  }
  __syncthreads();
  SS_MARK(9);

  // ---- C.2: Adam on everything this workgroup owns
  // layer 2: rows c0 .. c0 + 15 (registers of B.4), bias entries
  {
    const int lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int cj = cg + 4 * t;
      if (32 * cj < H) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
          const int64_t e = p.w_off[1] + (int64_t)(c0 + row) * H + 32 * cj + l31;
          ss_adam_math(p, g_w1[t][i], ap[t][i], am[t][i], av[t][i]);
          p.params[e] = ap[t][i]; p.m[e] = am[t][i]; p.v[e] = av[t][i];
        }
      }
    }
  }
  if (tid < SS_COLS) ss_adam(p, g_b1, p.b_off[1] + c0 + tid);
  // head: columns c0 .. c0 + 15 of every output row; bias and log-std in workgroup 0
  if (tid < 8 * SS_COLS) {
    const int o = tid / SS_COLS, c = tid % SS_COLS;
    if (o < A) ss_adam(p, g_head, p.w_off[2] + (int64_t)o * H + c0 + c);
  }
  if (blockIdx.x == 0 && tid < 8 && tid < A) ss_adam(p, g_bh, p.b_off[2] + tid);
  if (blockIdx.x == 0 && tid == 64) ss_adam(p, p.learn_std ? dlogstd_s : 0.f, 0);
  // layer 1: rows c0 .. c0 + 15: dW1[n][k] = sum_r dZ1[r][n] X[r][k]
  for (int e = tid; e < SS_COLS * ld0; e += SS_THREADS) {
    const int n = e / ld0, k = e % ld0;
    if (k < in_w) {
      float g = 0.f;
      for (int rr = 0; rr < SS_ROWS; ++rr)
        g = fmaf(dzown[rr * SS_LDO + n], xs[rr * SS_LDX + k], g);
      ss_adam(p, g, p.w_off[0] + (int64_t)(c0 + n) * ld0 + k);
    }
  }
  if (tid >= 128 && tid < 128 + SS_COLS) {
    const int n = tid - 128;
    float g = 0.f;
    for (int rr = 0; rr < SS_ROWS; ++rr) g += dzown[rr * SS_LDO + n];
    ss_adam(p, g, p.b_off[0] + c0 + n);
  }
  __syncthreads();
  SS_MARK(10);
}

}  // namespace

// ---------------------------------------------------------------------------
// host side (called by update.cpp; not part of the C ABI)
// ---------------------------------------------------------------------------
extern "C" int ga_small_step_supported(int n_layers, const int* dims, int64_t M) {
  if (n_layers != 3) return 0;
  const int in_w = dims[0], H = dims[1], out_w = dims[3];
  return dims[2] == H && H % 32 == 0 && H >= 32 && H <= SS_HMAX && in_w >= 1 &&
         in_w <= 32 && out_w >= 1 && out_w <= 8 && M >= 1 && M <= SS_ROWS;
}

// Residency: the grid barriers need all H / 16 workgroups of a launch (and of the
// other chain's concurrent launch) on the device at once.  A plain launch of a
// grid the device can hold becomes resident whatever else is queued (nothing that
// runs waits for it), so the check is grid x concurrent <= CUs x the occupancy
// query's workgroups per CU -- one per CU at 152 KB of LDS -- with the query's
// known over-count of one (MI355X_MICROARCH.md, residency) taken off when it
// reports more than one.  Shapes that fail it take the per-layer path.
static int g_resident_cap = -1;  // workgroups the device holds; -1 = not asked yet
static int g_resident_cap_forced = -1;
extern "C" int ga_set_small_step_resident_cap(int workgroups) {
  g_resident_cap_forced = workgroups;  // tests: 0 forces the per-layer path; < 0 = ask
  return 0;
}
extern "C" int ga_small_step_resident(int H, int concurrent) {
  if (g_resident_cap < 0) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) !=
            hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, small_step_kernel,
                                                     SS_THREADS, 0) != hipSuccess) {
      (void)hipGetLastError();
      g_resident_cap = 0;
    } else {
      if (per_cu > 1) --per_cu;
      g_resident_cap = cus * per_cu;
    }
  }
  const int cap = g_resident_cap_forced >= 0 ? g_resident_cap_forced : g_resident_cap;
  return (H / SS_COLS) * concurrent <= cap;
}

static int g_max_polls = 1 << 22;
// tests: 0 makes every waiter give up at once, i.e. forces the abort path
extern "C" int ga_set_small_step_max_polls(int polls) {
  g_max_polls = polls < 0 ? (1 << 22) : polls;
  return 0;
}

static int64_t g_launches = 0;
static long long* g_dbg = nullptr;
// developer hook: phase timestamps (100 MHz wall clock) of the most recent launch
extern "C" int ga_small_step_debug(long long* host_out16) {
  if (!g_dbg) {
    if (hipMalloc(&g_dbg, 16 * sizeof(long long)) != hipSuccess) return -1;
    (void)hipMemset(g_dbg, 0, 16 * sizeof(long long));
    return 1;  // armed: timestamps are recorded from the next launch on
  }
  (void)hipDeviceSynchronize();
  return hipMemcpy(host_out16, g_dbg, 16 * sizeof(long long), hipMemcpyDeviceToHost) ==
                 hipSuccess ? 0 : -1;
}
extern "C" int64_t ga_small_step_launches(void) { return g_launches; }

extern "C" int ga_small_step(const ga_small_step_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GA_REQUIRE(a && a->params && a->exp_avg && a->exp_avg_sq && a->X && a->xh2 && a->xdz &&
                 a->bar && a->loss_out && a->fault,
             "ga_small_step: null pointer");
  const int dims[4] = {a->in_w, a->H, a->H, a->out_w};
  GA_REQUIRE(ga_small_step_supported(3, dims, a->M), "ga_small_step: unsupported shape");
  GA_REQUIRE(ga_aligned16(a->X) && a->ldx % 4 == 0 && a->ldx >= ((a->in_w + 3) & ~3),
             "ga_small_step: X rows must be 16-B aligned quads");
  GA_REQUIRE(ga_aligned16(a->params) && ga_aligned16(a->xh2) && ga_aligned16(a->xdz) &&
                 a->w_off[0] % 4 == 0 && a->w_off[1] % 4 == 0 && a->w_off[2] % 4 == 0,
             "ga_small_step: alignment");
  GA_REQUIRE(a->kind != 1 ? (a->actions && a->adv && (a->algo == 1 || a->old_ll))
                          : (a->returns != nullptr),
             "ga_small_step: missing minibatch arrays");
  SmallStepParams p;
  p.params = a->params; p.m = a->exp_avg; p.v = a->exp_avg_sq;
  for (int i = 0; i < 3; ++i) { p.w_off[i] = a->w_off[i]; p.b_off[i] = a->b_off[i]; }
  p.in_w = a->in_w; p.H = a->H; p.out_w = a->out_w; p.M = a->M;
  p.X = a->X; p.ldx = a->ldx; p.idx = a->idx; p.kind = a->kind;
  p.double_softmax = a->double_softmax;
  p.actions = a->actions; p.lda = a->lda; p.old_ll = a->old_ll; p.adv = a->adv;
  p.returns = a->returns; p.algo = a->algo; p.clip = a->clip;
  p.has_min = a->has_min; p.has_max = a->has_max; p.min_log_std = a->min_log_std;
  p.max_log_std = a->max_log_std;
  p.ent_coeff = a->ent_coeff; p.ent_regularized = a->ent_flags & 1;
  p.ent_softplus = (a->ent_flags >> 1) & 1; p.ent_stop_grad = (a->ent_flags >> 2) & 1;
  p.lerp_w = (float)(1.0 - a->beta1);
  p.beta2 = (float)a->beta2;
  p.one_minus_beta2 = (float)(1.0 - a->beta2);
  const double bc1 = 1.0 - pow(a->beta1, (double)a->step);
  const double bc2 = 1.0 - pow(a->beta2, (double)a->step);
  p.neg_step_size = (float)(-(a->lr / bc1));
  p.bc2_sqrt = (float)sqrt(bc2);
  p.eps = (float)a->eps;
  p.learn_std = a->learn_std;
  p.dbg = g_dbg;
  p.max_polls = g_max_polls;
  p.xh2 = a->xh2; p.xdz = a->xdz; p.bar = a->bar; p.loss_out = a->loss_out; p.fault = a->fault;
  hipLaunchKernelGGL(small_step_kernel, dim3((unsigned)(a->H / SS_COLS)),
                     dim3(SS_THREADS), 0, stream, p);
  GA_CHECK_LAUNCH("small_step");
  ++g_launches;
  return GA_OK;
}
