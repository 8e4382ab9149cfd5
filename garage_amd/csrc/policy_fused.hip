// Fused rollout step: MLP policy forward + action head in ONE launch.
//
// Replaces, per vectorised step of VecWorker.step_episode
// (sampler/vec_worker.py:176-204), the chain
//   StochasticPolicy.get_actions -> GaussianMLPModule.forward -> MLP layers ->
//   dist.sample()  (torch/policies/stochastic_policy.py:46-89,
//   torch/modules/gaussian_mlp_module.py:158-192, multi_headed_mlp_module.py:136-151)
// and the per-env appends of observations / actions / agent_info.
//
// One workgroup (4 waves) owns 16 envs for the whole network (n = 4096: 256
// workgroups, one per CU): the activations of those rows never leave the CU (two
// [16][H+4] fp32 tiles in LDS, ping / pong between layers), the weights stream from
// L2 through a double-buffered [N][32+4] LDS stage in 32-wide k chunks (16-B loads,
// branch free), hidden layers run on v_mfma_f32_16x16x4_f32 (wave w owns the
// 16-column tiles w, w + 4, ...; means agree with the per-layer path to rounding),
// the narrow output layer is a 16-lane VALU dot product, and the head (Gaussian:
// mean + std * noise; categorical: inverse CDF) writes the action and the rollout
// buffers.  Hidden widths up to 256 (C2, C3); wider nets use the per-layer path.
// With the synthetic env the thread that sampled an env's action also steps it, and
// a whole rollout is ONE launch with the weights resident on the CU (see the
// kernel).
#include "common.h"
#include "prof.h"

#include "rollout_dev.h"

// ---- Audit (round 3) of out-of-range lanes / idle waves.  Clamped loads (value
// discarded by the matching store's mask): WeightStage::load and the training
// forward's wload -- nrow = min(f >> 3, N - 1), k = min(k0 + 4 (f & 7), last vector of
// the row).  Guarded loads: the register-resident second layer (ncol < N && k < K,
// a 16-B read at k <= ld - 4), biases (ncol < dims[l + 1], tid < dims[L]), the output
// layer's [N][ld] block (e < N * ld / 4), observations (env < n && c < in_w), noise
// rows (per env < n).  Nothing is fetched from an index derived from a wave number
// alone.
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 16;          // envs per workgroup (one 16 x 16 MFMA tile of rows)
constexpr int HMAX = 256;         // widest supported layer
constexpr int LDACT = HMAX + 4;   // activation tile row stride (floats)
constexpr int KC = 32;            // k chunk
constexpr int LDW = KC + 4;       // weight stage row stride
constexpr int MAX_OUT = 32;       // widest output head

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                            uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float unit_interval(uint32_t u) {
  return ((float)(u >> 8) + 0.5f) * 5.9604644775390625e-08f;
}
__device__ __forceinline__ void box_muller(uint32_t u0, uint32_t u1, float* z0,
                                           float* z1) {
  const float a = unit_interval(u0), b = unit_interval(u1);
  const float rad = sqrtf(-2.f * logf(a));
  float s, c;
  sincosf(6.28318530717958647692f * b, &s, &c);
  *z0 = rad * c;
  *z1 = rad * s;
}
__device__ __forceinline__ float tanh_fast(float x) {
  return ga_tanh(x);  // common.h
}

struct FusedParams {
  int n_layers;
  int dims[9];
  int64_t w_off[8], b_off[8];
  const float* params;
  int64_t n, env_id0;
  int kind;  // 0 gaussian, 1 categorical
  int has_min, has_max;
  float min_log_std, max_log_std;
  const float* noise;
  int64_t ldn;
  uint32_t k0, k1, step;
  int double_softmax;
  const float* obs;
  int64_t ldo;
  int64_t col, Tcap;
  float* action;
  int64_t lda;
  float* obs_buf;
  float* act_buf;
  float* head_buf;
  int64_t ldh;
  // env_step: the synthetic env's step, the NormalizedEnv statistics, the
  // bookkeeping and the reset of finished envs follow in the same launch, each env by
  // the thread that sampled its action (rollout_dev.h: the code of
  // synth_step_record_kernel)
  int env_step;
  ga_rollout::EnvStepArgs es;
  // n_steps > 1 (needs env_step): the workgroup takes its envs through n_steps
  // consecutive rollout steps in this one launch -- nothing couples the envs of
  // different workgroups within a rollout -- alternating between the two
  // observation buffers (obs / es.seen_next and, with observation normalisation,
  // es.raw_obs / es.raw_next)
  int n_steps;
  long long* dbg;  // developer hook: phase timestamps of workgroup 0
};

#define PS_STAMP(i) \
  if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0 && sidx == p.n_steps / 2) \
    p.dbg[(i) + (RES ? 16 : 0)] = wall_clock64()

// Stage W[:, k0 : k0 + 32] of a [N][ldw] weight matrix: registers -> LDS.
struct WeightStage {
  float4 regs[HMAX * KC / 4 / 256];  // 8 vectors per thread at N = 256

  __device__ __forceinline__ void load(const float* __restrict__ W, int ldw, int N,
                                       int K, int k0) {
    const int last = max(((K + 3) & ~3) - 4, 0);
#pragma unroll
    for (int i = 0; i < HMAX * KC / 4 / 256; ++i) {
      const int f = threadIdx.x + 256 * i;
      const int nrow = min(f >> 3, N - 1);
      const int k = min(k0 + 4 * (f & 7), last);
      regs[i] = *reinterpret_cast<const float4*>(W + (int64_t)nrow * ldw + k);
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ stage, int N, int K,
                                        int k0) const {
#pragma unroll
    for (int i = 0; i < HMAX * KC / 4 / 256; ++i) {
      const int f = threadIdx.x + 256 * i;
      const int nrow = f >> 3;
      const int k = 4 * (f & 7);
      float4 v = regs[i];
      const bool ok = nrow < N;
      v.x = (ok && k0 + k + 0 < K) ? v.x : 0.f;
      v.y = (ok && k0 + k + 1 < K) ? v.y : 0.f;
      v.z = (ok && k0 + k + 2 < K) ? v.z : 0.f;
      v.w = (ok && k0 + k + 3 < K) ? v.w : 0.f;
      *reinterpret_cast<float4*>(stage + nrow * LDW + k) = v;
    }
  }
};

// Butterfly sum over 16 consecutive lanes (xor 1, 2, 4, 8) on DPP lane permutes:
// after the first two stages a quad holds one value, after the third a half row, so
// the mirrored (half) row supplies what lane ^ 4 / lane ^ 8 holds -- the same
// additions, in the same order, as four shuffles.
__device__ __forceinline__ float sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
           __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
           __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
           __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
           __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// The hidden layers run on v_mfma_f32_16x16x4_f32 (lane l: A[row l % 16][slot l / 16],
// B[slot l / 16][col l % 16], D[row 4 (l / 16) + reg][col l % 16]); wave w owns the
// 16-column tiles w, w + 4, w + 8, w + 12 of a layer's outputs.  Within a 16-deep
// group MFMA q gives slot kq the element k = 16 G + 4 kq + q, so A and B fragments
// are one 16-B read per lane and group.
constexpr int TPW = HMAX / 64;  // tiles per wave

// One 32-deep chunk with the B fragments in a staged [N][LDW] chunk.
//   A: act + (l % 16) * LDACT + 32 * chunk + 4 * (l / 16)
//   B: stage + (16 * wave + l % 16) * LDW + 4 * (l / 16)
template <int NT>
__device__ __forceinline__ void staged_chunk(const float* __restrict__ A,
                                             const float* __restrict__ B,
                                             f32x4 (&acc)[TPW]) {
#pragma unroll
  for (int G = 0; G < KC / 16; ++G) {
    const float4 av = *reinterpret_cast<const float4*>(A + 16 * G);
    const float a4[4] = {av.x, av.y, av.z, av.w};
    float b4[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 bv = *reinterpret_cast<const float4*>(B + 64 * t * LDW + 16 * G);
      b4[t][0] = bv.x; b4[t][1] = bv.y; b4[t][2] = bv.z; b4[t][3] = bv.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q], b4[t][q], acc[t], 0, 0, 0);
  }
}

// NG groups of 16 k of a hidden layer whose B operands sit in registers
// (wreg[t][4 G + q] = W[16 (wave + 4 t) + l % 16][16 G + 4 (l / 16) + q]).
template <int NG, int NT>
__device__ __forceinline__ void resident_layer(const float* __restrict__ A,
                                               const float (&wreg)[TPW][HMAX / 4],
                                               f32x4 (&acc)[TPW]) {
#pragma unroll
  for (int G = 0; G < NG; ++G) {
    const float4 av = *reinterpret_cast<const float4*>(A + 16 * G);
    const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q], wreg[t][4 * G + q], acc[t],
                                                      0, 0, 0);
  }
}
template <int NT>
__device__ __forceinline__ void resident_layer_k(const float* __restrict__ A, int K,
                                                 const float (&wreg)[TPW][HMAX / 4],
                                                 f32x4 (&acc)[TPW]) {
  // (straight-line per depth: a branch per k group would make the compiler move the
  // accumulators at every join; the tiles are zero beyond K)
  if (K <= 64) resident_layer<4, NT>(A, wreg, acc);
  else if (K <= 128) resident_layer<8, NT>(A, wreg, acc);
  else resident_layer<16, NT>(A, wreg, acc);
}

// RES (a whole rollout in one launch, observations no wider than one k chunk, one
// or two hidden layers): the weights stay on the CU for all the steps -- the first
// layer's chunk in wst[0], the output layer's rows in wst[1], and the second hidden
// layer's [N][K] matrix in REGISTERS (each lane holds the 4 x 64 B operands its
// MFMAs consume: 256 of the 512 registers a wave has at one wave per SIMD), so that
// layer runs without a barrier or a weight fetch.  Same k order per accumulator as
// the streamed loop: bit-identical.
template <bool RES>
__global__ __launch_bounds__(256) void policy_step_fused_kernel(FusedParams p) {
  __shared__ __attribute__((aligned(16))) float act[2][ROWS * LDACT];
  __shared__ __attribute__((aligned(16))) float wst[2][HMAX * LDW];
  __shared__ float head[ROWS][MAX_OUT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int L = p.n_layers;
  float wreg[TPW][HMAX / 4];
  float bias_r[2][TPW];  // RES: hidden biases of this lane's columns
#pragma unroll
  for (int t = 0; t < TPW; ++t) bias_r[0][t] = bias_r[1][t] = 0.f;
  __shared__ float obias[MAX_OUT];
  if constexpr (RES) {
    for (int l = 0; l < L - 1; ++l)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int ncol = 16 * (wave + 4 * t) + r16;
        const float b = ncol < p.dims[l + 1] ? p.params[p.b_off[l] + ncol] : 0.f;
        if (l == 0) bias_r[0][t] = b;
        else bias_r[1][t] = b;
      }
    if (tid < p.dims[L]) obias[tid] = p.params[p.b_off[L - 1] + tid];
    // (the register-resident layer multiplies whole tiles: no stale columns)
    for (int e = tid; e < ROWS * LDACT; e += 256) act[0][e] = act[1][e] = 0.f;
    {
      const int K = p.dims[0], N = p.dims[1];
      WeightStage ws;
      ws.load(p.params + p.w_off[0], (K + 3) & ~3, N, K, 0);
      ws.store(wst[0], N, K, 0);
    }
    if (L == 3) {
      const int K = p.dims[1], N = p.dims[2];
      const int ldw = (K + 3) & ~3;
      const float* W = p.params + p.w_off[1];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int ncol = 16 * (wave + 4 * t) + r16;
#pragma unroll
        for (int G = 0; G < HMAX / 16; ++G) {
          const int k = 16 * G + 4 * kq;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ncol < N && k < K) {
            v = *reinterpret_cast<const float4*>(W + (int64_t)ncol * ldw + k);
            if (k + 1 >= K) v.y = 0.f;
            if (k + 2 >= K) v.z = 0.f;
            if (k + 3 >= K) v.w = 0.f;
          }
          wreg[t][4 * G + 0] = v.x; wreg[t][4 * G + 1] = v.y;
          wreg[t][4 * G + 2] = v.z; wreg[t][4 * G + 3] = v.w;
        }
      }
    }
    {
      const int K = p.dims[L - 1], N = p.dims[L];
      const int ldw = (K + 3) & ~3;
      const float* W = p.params + p.w_off[L - 1];
      for (int e = tid; e < N * (ldw / 4); e += 256)
        reinterpret_cast<float4*>(wst[1])[e] = reinterpret_cast<const float4*>(W)[e];
    }
    __syncthreads();
  }
  for (int sidx = 0; sidx < p.n_steps; ++sidx) {
  // this step's column, Philox counter and observation buffers (they swap roles
  // every step)
  const int64_t col = p.col + sidx;
  const uint32_t step = p.step + (uint32_t)sidx;
  const bool odd = sidx & 1;
  const float* obs = odd ? p.es.seen_next : p.obs;
  ga_rollout::EnvStepArgs es = p.es;
  if (p.env_step) {
    es.p.col = col;
    es.seen_next = odd ? const_cast<float*>(p.obs) : p.es.seen_next;
    es.p.next_obs = es.seen_next;
    if (p.es.raw_next != p.es.seen_next) {  // NormalizedEnv: the env's own rows
      es.raw_obs = odd ? p.es.raw_next : p.es.raw_obs;
      es.raw_next = odd ? const_cast<float*>(p.es.raw_obs) : p.es.raw_next;
    } else {
      es.raw_obs = obs;
      es.raw_next = es.seen_next;
    }
  }

  // ---- observations -> act[0] (zero padded to a multiple of the k chunk) and
  //      into the rollout buffer (the list append of vec_worker.py:188)
  PS_STAMP(0);
  const int in_w = p.dims[0];
  const int in_pad = (in_w + KC - 1) / KC * KC;
  for (int e = tid; e < ROWS * in_pad; e += 256) {
    const int r = e / in_pad, c = e % in_pad;
    const int64_t env = row0 + r;
    float v = 0.f;
    if (env < p.n && c < in_w) {
      v = obs[env * p.ldo + c];
      p.obs_buf[(env * p.Tcap + col) * p.ldo + c] = v;
    }
    act[0][r * LDACT + c] = v;
  }
  // the env threads fetch what their env's step will read now, behind the network
  ga_rollout::EnvPre pre;
  if (p.env_step && tid < ROWS && row0 + tid < p.n)
    pre = ga_rollout::env_prefetch(es, row0 + tid);
  __syncthreads();
  PS_STAMP(1);

  // ---- hidden layers on the matrix cores
  int cur = 0;
  for (int l = 0; l < L - 1; ++l) {
    const int K = p.dims[l], N = p.dims[l + 1];
    const int ldw = (K + 3) & ~3;
    const float* W = p.params + p.w_off[l];
    const float* bias = p.params + p.b_off[l];
    const int nk = (K + KC - 1) / KC;
    const int n_pad = (N + KC - 1) / KC * KC;
    // this wave's tiles: columns 16 (wave + 4 t); a narrow layer is one tile per wave
    const bool wave_on = 16 * wave < n_pad;
    const bool one_tile = n_pad <= 64;
    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    const float* Arow = act[cur] + r16 * LDACT + 4 * kq;
    if constexpr (RES) {
      if (wave_on) {
        if (l == 0) {
          const float* B = wst[0] + (16 * wave + r16) * LDW + 4 * kq;
          if (one_tile) staged_chunk<1>(Arow, B, acc);
          else staged_chunk<TPW>(Arow, B, acc);
        } else {
          if (one_tile) resident_layer_k<1>(Arow, K, wreg, acc);
          else resident_layer_k<TPW>(Arow, K, wreg, acc);
        }
      }
    } else {
    WeightStage ws;
    ws.load(W, ldw, N, K, 0);
    ws.store(wst[0], N, K, 0);
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
      const bool more = s + 1 < nk;
      if (more) ws.load(W, ldw, N, K, (s + 1) * KC);
      if (wave_on) {
        const float* B = wst[s & 1] + (16 * wave + r16) * LDW + 4 * kq;
        if (one_tile) staged_chunk<1>(Arow + s * KC, B, acc);
        else staged_chunk<TPW>(Arow + s * KC, B, acc);
      }
      if (more) ws.store(wst[(s + 1) & 1], N, K, (s + 1) * KC);
      __syncthreads();
    }
    }  // streamed weights
    (void)nk;
    // bias + tanh -> the other activation tile (zero padded to the k chunk)
    float* out = act[cur ^ 1];
    if (wave_on) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (t == 0 || !one_tile) {
          const int ncol = 16 * (wave + 4 * t) + r16;
          const float bv = RES ? (l == 0 ? bias_r[0][t] : bias_r[1][t])
                               : (ncol < N ? bias[ncol] : 0.f);
          // (straight-line: tanh of every element, then one guarded run of stores)
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float th = tanh_fast(acc[t][r] + bv);
            v[r] = ncol < N ? th : 0.f;
          }
          if (ncol < n_pad) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(4 * kq + r) * LDACT + ncol] = v[r];
          }
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    PS_STAMP(2 + l);
  }

  // ---- narrow output layer: its weights go to LDS once; 16 lanes per row hold
  //      their k slice of the row in registers and dot it with every output
  {
    const int K = p.dims[L - 1], N = p.dims[L];
    const int ldw = (K + 3) & ~3;
    const float* W = p.params + p.w_off[L - 1];
    const float* bias = p.params + p.b_off[L - 1];
    float* wo = wst[RES ? 1 : 0];  // [N][ldw]: N <= 32, ldw <= 256 -> fits one stage
    if constexpr (!RES)
      for (int e = tid; e < N * (ldw / 4); e += 256)
        reinterpret_cast<float4*>(wo)[e] = reinterpret_cast<const float4*>(W)[e];
    const int r = tid >> 4, part = tid & 15;
    const float* a = act[cur] + r * LDACT;
    // (branch free: out-of-range slices read slice 0 and are selected away, so the
    // LDS reads of a row go out together instead of one latency after another)
    float4 xr[HMAX / 64];
#pragma unroll
    for (int i = 0; i < HMAX / 64; ++i) {
      const int k = part * 4 + 64 * i;
      float4 v = *reinterpret_cast<const float4*>(a + (k < K ? k : 0));
      v.x = k < K ? v.x : 0.f;
      v.y = k + 1 < K ? v.y : 0.f;
      v.z = k + 2 < K ? v.z : 0.f;
      v.w = k + 3 < K ? v.w : 0.f;
      xr[i] = v;
    }
    __syncthreads();
    for (int o = 0; o < N; ++o) {
      const float* w = wo + o * ldw + part * 4;
      float4 wv[HMAX / 64];
#pragma unroll
      for (int i = 0; i < HMAX / 64; ++i)
        wv[i] = *reinterpret_cast<const float4*>(
            w + (part * 4 + 64 * i < ldw ? 64 * i : 0));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < HMAX / 64; ++i) {
        const float t = sum + (xr[i].x * wv[i].x + xr[i].y * wv[i].y +
                               xr[i].z * wv[i].z + xr[i].w * wv[i].w);
        sum = part * 4 + 64 * i < ldw ? t : sum;
      }
      sum = sum16(sum);
      if (part == 0) head[r][o] = sum + (RES ? obias[o] : bias[o]);
    }
  }
  __syncthreads();
  PS_STAMP(10);

  // ---- action head: one thread per env
  int ended_len = 0;
  if (tid < ROWS) {
    const int64_t env = row0 + tid;
    if (env < p.n) {
      const int N = p.dims[L];
      const int64_t cell = env * p.Tcap + col;
      const float* h = head[tid];
      float* act_row = head[tid];  // the sampled action replaces the mean / scores
      if (p.head_buf) {
        // agent_info: Gaussian mean (probabilities are written below)
        if (p.kind == 0)
          for (int j = 0; j < N; ++j) p.head_buf[cell * p.ldh + j] = h[j];
      }
      if (p.kind == 0) {
        const float s = ga_log_std(p.params[0], p.has_min, p.min_log_std, p.has_max,
                                   p.max_log_std, nullptr);
        const float sd = expf(s);
        for (int b = 0; b * 4 < N; ++b) {
          float z[4];
          if (p.noise) {
            for (int j = 0; j < 4 && b * 4 + j < N; ++j)
              z[j] = p.noise[env * p.ldn + b * 4 + j];
          } else {
            const U4 rr = philox4x32_10((uint32_t)(p.env_id0 + env), step,
                                        (uint32_t)b, 3u << 16, p.k0, p.k1);
            box_muller(rr.x, rr.y, &z[0], &z[1]);
            box_muller(rr.z, rr.w, &z[2], &z[3]);
          }
          for (int j = 0; j < 4 && b * 4 + j < N; ++j) {
            const float a = h[b * 4 + j] + sd * z[j];
            p.action[env * p.lda + b * 4 + j] = a;
            p.act_buf[cell * p.lda + b * 4 + j] = a;
            act_row[b * 4 + j] = a;
          }
        }
      } else {
        float mx = h[0];
        for (int j = 1; j < N; ++j) mx = fmaxf(mx, h[j]);
        float den = 0.f;
        for (int j = 0; j < N; ++j) den += expf(h[j] - mx);
        float den2 = 0.f;
        if (p.double_softmax)
          for (int j = 0; j < N; ++j) den2 += expf(expf(h[j] - mx) / den);
        float u;
        if (p.noise) {
          u = p.noise[env * p.ldn];
        } else {
          const U4 rr = philox4x32_10((uint32_t)(p.env_id0 + env), step, 0u,
                                      3u << 16, p.k0, p.k1);
          u = unit_interval(rr.x);
        }
        float cdf = 0.f;
        int pick = N - 1;
        bool found = false;
        for (int j = 0; j < N; ++j) {
          float pr = expf(h[j] - mx) / den;
          if (p.double_softmax) pr = expf(pr) / den2;
          if (p.head_buf) p.head_buf[cell * p.ldh + j] = pr;
          cdf += pr;
          if (!found && u < cdf) { pick = j; found = true; }
        }
        p.action[env * p.lda] = (float)pick;
        p.act_buf[cell * p.lda] = (float)pick;
        act_row[0] = (float)pick;
      }
      if (p.env_step) ended_len = ga_rollout::env_step_one(es, env, pre, act_row);
    }
  }
  // (episode counts of the step: a wave-aggregated integer atomic; the envs of a
  // workgroup all sit in wave 0)
  if (p.env_step && wave == 0) ga_rollout::record_counts(es.p, ended_len);
  PS_STAMP(11);
  // the next step reads what the env threads just wrote (same workgroup: one CU,
  // one L1) and reuses the LDS tiles
  if (sidx + 1 < p.n_steps) __syncthreads();
  }  // steps
}

// ---- the same network, training forward ---------------------------------------
// All layers of one minibatch forward in one launch: 8 waves own 32 (gathered)
// rows; hidden activations go to LDS for the next layer AND to HBM for the
// backward pass; the narrow output layer is a 16-lane VALU dot.  Replaces the
// three per-layer GEMM launches of ga_mlp_forward_f32 (the activations' HBM
// read between layers disappears, the weights come from L2).
struct TrainFwdParams {
  int n_layers;
  int dims[9];
  int64_t w_off[8], b_off[8], act_off[8];
  const float* params;
  const float* X;
  int64_t ldx;
  const int32_t* idx;
  int64_t M;
  float* acts;
  float* out;
  int64_t ldo;
};

constexpr int TROWS = 64;  // rows per workgroup of the training forward

__global__ __launch_bounds__(512) void mlp_train_fwd_fused_kernel(TrainFwdParams p) {
  // 64 rows x 8 waves: wave (ri, cj) owns rows [32 ri, 32 ri + 32) and columns
  // [64 cj, 64 cj + 64).  One activation tile, updated IN PLACE: a layer's
  // outputs wait in the accumulators until every wave has finished reading the
  // inputs, so LDS holds 64 rows (66.5 KB) + two weight stages (74 KB) and each
  // weight chunk fetched from L2 is amortised over 64 rows (32 MFMAs per wave per
  // chunk = ~1.7 us per SIMD, enough to cover the fetch of the next chunk).
  constexpr int NT = 512;
  __shared__ __attribute__((aligned(16))) float act[TROWS * LDACT];
  __shared__ __attribute__((aligned(16))) float wst[2][HMAX * LDW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int ri = wave >> 2, cj = wave & 3;
  const int64_t row0 = (int64_t)blockIdx.x * TROWS;
  const int L = p.n_layers;

  const int in_w = p.dims[0];
  const int in_pad = (in_w + KC - 1) / KC * KC;
  for (int e = tid; e < TROWS * in_pad; e += NT) {
    const int r = e / in_pad, c = e % in_pad;
    const int64_t m = row0 + r;
    float v = 0.f;
    if (m < p.M && c < in_w) {
      const int64_t src = p.idx ? (int64_t)p.idx[m] : m;
      v = p.X[src * p.ldx + c];
    }
    act[r * LDACT + c] = v;
  }
  __syncthreads();

  for (int l = 0; l < L - 1; ++l) {
    const int K = p.dims[l], N = p.dims[l + 1];
    const int ldw = (K + 3) & ~3;
    const int ldh = (N + 3) & ~3;
    const float* W = p.params + p.w_off[l];
    const float* bias = p.params + p.b_off[l];
    float* gact = p.acts + p.act_off[l];
    const int nk = (K + KC - 1) / KC;
    const int n0 = cj * 64;
    const bool wave_on = n0 < N;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float4 wr[4];  // weight stage: 256 x 32 floats = 2048 vectors, 4 per thread
    const int last = max(((K + 3) & ~3) - 4, 0);
    auto wload = [&](int k0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + NT * i;
        const int nrow = min(f >> 3, N - 1);
        wr[i] = *reinterpret_cast<const float4*>(W + (int64_t)nrow * ldw +
                                                 min(k0 + 4 * (f & 7), last));
      }
    };
    auto wstore = [&](float* stage, int k0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + NT * i;
        const int nrow = f >> 3, k = 4 * (f & 7);
        float4 v = wr[i];
        const bool ok = nrow < N;
        v.x = (ok && k0 + k + 0 < K) ? v.x : 0.f;
        v.y = (ok && k0 + k + 1 < K) ? v.y : 0.f;
        v.z = (ok && k0 + k + 2 < K) ? v.z : 0.f;
        v.w = (ok && k0 + k + 3 < K) ? v.w : 0.f;
        *reinterpret_cast<float4*>(stage + nrow * LDW + k) = v;
      }
    };
    wload(0);
    wstore(wst[0], 0);
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
      const bool more = s + 1 < nk;
      if (more) wload((s + 1) * KC);
      if (wave_on) {
        const float* A = act + (32 * ri + l31) * LDACT + s * KC;
        const float* B = wst[s & 1] + (n0 + l31) * LDW;
#pragma unroll
        for (int g = 0; g < KC / 8; ++g) {
          const float4 av = *reinterpret_cast<const float4*>(A + 8 * g + 4 * half);
          const float4 b0 = *reinterpret_cast<const float4*>(B + 8 * g + 4 * half);
          const float4 b1 =
              *reinterpret_cast<const float4*>(B + 32 * LDW + 8 * g + 4 * half);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc[1], 0, 0, 0);
        }
      }
      if (more) wstore(wst[(s + 1) & 1], (s + 1) * KC);
      __syncthreads();
    }
    // every wave is past its last read of `act`: overwrite it with this layer
    const int n_pad = (N + KC - 1) / KC * KC;
    if (wave_on) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ncol = n0 + 32 * j + l31;
        const float bv = ncol < N ? bias[ncol] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * ri + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = ncol < N ? tanh_fast(acc[j][r] + bv) : 0.f;
          if (ncol < n_pad) act[m * LDACT + ncol] = v;
          if (ncol < N && row0 + m < p.M) gact[(row0 + m) * ldh + ncol] = v;
        }
      }
    }
    __syncthreads();
  }

  // narrow output layer: its weights go to LDS once, then 8 lanes per row
  {
    const int K = p.dims[L - 1], N = p.dims[L];
    const int ldw = (K + 3) & ~3;
    const float* W = p.params + p.w_off[L - 1];
    const float* bias = p.params + p.b_off[L - 1];
    float* wo = wst[0];  // [N][ldw] fits: N <= 32, ldw <= 256
    for (int e = tid; e < N * (ldw / 4); e += NT)
      reinterpret_cast<float4*>(wo)[e] = reinterpret_cast<const float4*>(W)[e];
    __syncthreads();
    const int r = tid >> 3, part = tid & 7;
    const float* a = act + r * LDACT;
    for (int o = 0; o < N; ++o) {
      const float* w = wo + o * ldw;
      float sum = 0.f;
      for (int k = part * 4; k < K; k += 32) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
        const float4 xv = *reinterpret_cast<const float4*>(a + k);
        sum += xv.x * wv.x;
        if (k + 1 < K) sum += xv.y * wv.y;
        if (k + 2 < K) sum += xv.z * wv.z;
        if (k + 3 < K) sum += xv.w * wv.w;
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      if (part == 0 && row0 + r < p.M) p.out[(row0 + r) * p.ldo + o] = sum + bias[o];
    }
  }
}

}  // namespace

struct ga_mlp_desc_c {
  int32_t n_layers;
  int32_t dims[9];
  int64_t w_off[8];
  int64_t b_off[8];
  int64_t act_off[8];
  int32_t hidden_act;  // 0 tanh, 1 relu, 2 none: these kernels implement tanh
  int32_t output_act;  // 0 none, 1 tanh, 2 relu: ... and a linear output layer
  int32_t layer_norm;  // ... and no layer normalisation
  int32_t pad_;
  int64_t ln_off[8], lnx_off[8], lns_off[8];
};

struct ga_head_args_c {
  int64_t n, env_id0;
  int32_t A, kind;
  const float* head; int64_t ldh;
  const float* log_std; int32_t has_min, has_max; float min_log_std, max_log_std;
  const float* noise; int64_t ldn;
  uint64_t seed; uint32_t step; int32_t double_softmax;
  const float* obs; int64_t ldo; int32_t obs_dim;
  int64_t col, Tcap;
  float* action; int64_t lda;
  float* obs_buf; float* act_buf; float* head_buf;
};

// 1 when ga_policy_step_fused_f32 supports this network shape.
extern "C" int ga_policy_step_fused_supported(const ga_mlp_desc_c* d) {
  if (!d || d->n_layers < 1 || d->n_layers > 8) return 0;
  // tanh hidden layers, a linear output layer, no layer normalisation
  if (d->hidden_act != 0 || d->output_act != 0 || d->layer_norm) return 0;
  for (int l = 0; l < d->n_layers; ++l)
    if (d->dims[l] > HMAX) return 0;  // every layer INPUT lives in an LDS tile
  if (d->dims[d->n_layers] > MAX_OUT) return 0;
  return 1;
}

// `head` of args is ignored (the means / scores stay on chip); everything else
// as in ga_policy_head_sample.
static int policy_step_launch(const ga_mlp_desc_c* d, const float* params,
                              const ga_head_args_c* a, const ga_rollout::EnvStepArgs* es,
                              int64_t n_steps, hipStream_t stream);

static bool g_ps_no_resident = getenv("GARAGE_AMD_ROLLOUT_RESIDENT") &&
                               atoi(getenv("GARAGE_AMD_ROLLOUT_RESIDENT")) == 0;
static long long* g_ps_dbg = nullptr;
// developer hook: phase timestamps (100 MHz wall clock) of workgroup 0 of the most
// recent fused rollout step -- first call arms it, second call reads 32 values back
// (0 start, 1 observations staged, 2 + l hidden layer l done, 10 output layer,
// 11 sampled + env stepped; + 16: the same of the resident-weights kernel; the middle step of the launch)
extern "C" int ga_policy_step_debug(long long* host_out32) {
  if (!g_ps_dbg) {
    if (hipMalloc(&g_ps_dbg, 32 * sizeof(long long)) != hipSuccess) return -1;
    (void)hipMemset(g_ps_dbg, 0, 32 * sizeof(long long));
    return 1;
  }
  (void)hipDeviceSynchronize();
  return hipMemcpy(host_out32, g_ps_dbg, 32 * sizeof(long long), hipMemcpyDeviceToHost) ==
                 hipSuccess ? 0 : -1;
}

extern "C" int ga_policy_step_fused_f32(const ga_mlp_desc_c* d, const float* params,
                                        const ga_head_args_c* a, hipStream_t stream) {
  return policy_step_launch(d, params, a, nullptr, 1, stream);
}

// The same launch followed, per env, by the synthetic env's step + NormalizedEnv
// statistics + bookkeeping + reset (what ga_synth_env_step_record_norm does as its
// own launch), for n_steps consecutive rollout steps: every workgroup takes its 32
// envs through all of them (nothing couples envs within a rollout), alternating
// between `a->obs` and `rec->next_obs` (and the raw pair of `norm`).  `a->action` is
// what the env is stepped with.
extern "C" int ga_policy_env_step_fused_f32(const ga_mlp_desc_c* d, const float* params,
                                            const ga_head_args_c* a,
                                            const ga_synth_env* env,
                                            const ga_record_args* rec,
                                            const ga_norm_args* norm, int64_t n_steps,
                                            hipStream_t stream) {
  GA_REQUIRE(a && rec, "ga_policy_env_step_fused_f32: null pointer");
  GA_REQUIRE(n_steps >= 1 && a->col + n_steps <= a->Tcap,
             "ga_policy_env_step_fused_f32: steps exceed the rollout buffer");
  GA_REQUIRE(!a->noise || n_steps == 1,
             "ga_policy_env_step_fused_f32: teacher-forced noise is per step");
  ga_rollout::EnvStepArgs es;
  int rc = ga_build_env_step(env, rec, norm, a->action, a->lda, a->obs,
                             "ga_policy_env_step_fused_f32", &es);
  if (rc) return rc;
  GA_REQUIRE(es.e.n == a->n, "ga_policy_env_step_fused_f32: env count mismatch");
  GA_REQUIRE(es.p.col == a->col && es.p.col + n_steps <= es.p.Tcap,
             "ga_policy_env_step_fused_f32: record columns do not match the policy's");
  return policy_step_launch(d, params, a, &es, n_steps, stream);
}

static int policy_step_launch(const ga_mlp_desc_c* d, const float* params,
                              const ga_head_args_c* a, const ga_rollout::EnvStepArgs* es,
                              int64_t n_steps, hipStream_t stream) {
  GA_REQUIRE(d && params && a, "ga_policy_step_fused_f32: null pointer");
  GA_REQUIRE(ga_policy_step_fused_supported(d),
             "ga_policy_step_fused_f32: unsupported network shape");
  GA_REQUIRE(a->obs && a->action && a->obs_buf && a->act_buf,
             "ga_policy_step_fused_f32: null buffer");
  GA_REQUIRE(a->n > 0 && a->A == d->dims[d->n_layers] && a->obs_dim == d->dims[0],
             "ga_policy_step_fused_f32: head / network size mismatch");
  GA_REQUIRE(a->col >= 0 && a->col < a->Tcap,
             "ga_policy_step_fused_f32: col out of range");
  GA_REQUIRE(ga_aligned16(params), "ga_policy_step_fused_f32: params alignment");
  FusedParams p;
  p.n_layers = d->n_layers;
  for (int i = 0; i < 9; ++i) p.dims[i] = d->dims[i];
  for (int i = 0; i < 8; ++i) { p.w_off[i] = d->w_off[i]; p.b_off[i] = d->b_off[i]; }
  p.params = params; p.n = a->n; p.env_id0 = a->env_id0; p.kind = a->kind;
  p.has_min = a->has_min; p.has_max = a->has_max; p.min_log_std = a->min_log_std;
  p.max_log_std = a->max_log_std; p.noise = a->noise; p.ldn = a->ldn;
  p.k0 = (uint32_t)(a->seed & 0xffffffffu); p.k1 = (uint32_t)(a->seed >> 32);
  p.step = a->step; p.double_softmax = a->double_softmax; p.obs = a->obs;
  p.ldo = a->ldo; p.col = a->col; p.Tcap = a->Tcap; p.action = a->action;
  p.lda = a->lda; p.obs_buf = a->obs_buf; p.act_buf = a->act_buf;
  p.head_buf = a->head_buf; p.ldh = a->ldh;
  p.dbg = g_ps_dbg;
  p.env_step = es != nullptr;
  p.n_steps = (int)n_steps;
  if (es) p.es = *es;
  else memset(&p.es, 0, sizeof(p.es));
  const dim3 grid((unsigned)ga_ceil_div(a->n, ROWS));
  // a whole rollout in one launch keeps the weights on the CU (see the kernel)
  const bool resident = n_steps > 1 && d->dims[0] <= KC &&
                        (d->n_layers == 2 || d->n_layers == 3) && !g_ps_no_resident;
  if (resident) ga_prof_count(GA_PROF_ROLLOUT);
  if (resident)
    hipLaunchKernelGGL(policy_step_fused_kernel<true>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(policy_step_fused_kernel<false>, grid, dim3(256), 0, stream, p);
  GA_CHECK_LAUNCH("policy_step_fused");
  return GA_OK;
}

// Training / evaluation forward of the whole MLP in one launch (same contract
// as ga_mlp_forward_f32, which dispatches here when the shape is supported).
extern "C" int ga_mlp_forward_fused_f32(const ga_mlp_desc_c* d, const float* params,
                                        const float* X, int64_t ldx,
                                        const int32_t* row_idx, int64_t M,
                                        float* acts, float* out, int64_t ldo,
                                        hipStream_t stream) {
  GA_REQUIRE(d && params && X && out, "ga_mlp_forward_fused_f32: null pointer");
  GA_REQUIRE(ga_policy_step_fused_supported(d),
             "ga_mlp_forward_fused_f32: unsupported network shape");
  GA_REQUIRE(d->n_layers == 1 || acts, "ga_mlp_forward_fused_f32: acts needed");
  GA_REQUIRE(M > 0 && M < (1ll << 31), "ga_mlp_forward_fused_f32: bad M");
  GA_REQUIRE(ga_aligned16(params), "ga_mlp_forward_fused_f32: params alignment");
  TrainFwdParams p;
  p.n_layers = d->n_layers;
  for (int i = 0; i < 9; ++i) p.dims[i] = d->dims[i];
  for (int i = 0; i < 8; ++i) {
    p.w_off[i] = d->w_off[i]; p.b_off[i] = d->b_off[i]; p.act_off[i] = d->act_off[i];
  }
  p.params = params; p.X = X; p.ldx = ldx; p.idx = row_idx; p.M = M; p.acts = acts;
  p.out = out; p.ldo = ldo;
  hipLaunchKernelGGL(mlp_train_fwd_fused_kernel, dim3((unsigned)ga_ceil_div(M, TROWS)),
                     dim3(512), 0, stream, p);
  GA_CHECK_LAUNCH("mlp_train_fwd_fused");
  return GA_OK;
}
