// Wavefront-segmented reverse scan: discounted returns + GAE(lambda) advantages.
//
// Replaces, in one HBM pass,
//   * garage.np.discount_cumsum called row by row   (np/_functions.py:111-128,
//     call site torch/algos/vpg.py:149-153)
//   * garage.torch.compute_advantages               (torch/_functions.py:25-85)
// including the reference's zero-padding semantics for short episodes
// (SURVEY.md Q2): padded baseline cells hold v0 = V(0-obs), so an episode of
// length L < P starts its backward recursion from a closed-form carry C0(P-L)
// and bootstraps its last step from v0.
//
// Layout: a "row" is a contiguous run of fp32 steps -- one env's slice of the
// env-major (n_envs, T) rollout buffer (mode 0: episode ends are marked by
// tail[i] = episode length, 0 elsewhere), or one episode of a packed /
// padded batch (mode 1: the row's last element is the episode end).  A group of
// `lpr` lanes (power of two <= 64) owns a row; each lane owns 4 consecutive
// steps (one 16-B load per array), composes their affine maps right-to-left,
// and the group runs a log-step suffix scan over (decay, offset) pairs with
// wave shuffles.  Rows longer than 4*lpr are walked chunk by chunk from the
// right with a group-uniform carry.  Both recurrences run in fp64 registers
// (the reference does the returns in float64; HBM, not the VALU, bounds this
// kernel) and are stored as fp32.
//
// Algorithmic HBM bytes: 16 per step (r, V in; A, G out) (+2 for the tail flag).
#include "common.h"
#include <hip/hip_ext.h>

#include "prof.h"

namespace {

struct Pair {
  double d;  // decay applied to the carry coming from the right
  double x;  // value when that carry is 0
};

__device__ __forceinline__ double shfl_down_d(double v, int o, int width) {
  return __shfl_down(v, o, width);
}

struct ScanParams {
  const float* rew;
  const float* val;
  const float* bonus;      // optional per-step reward bonus (entropy 'max')
  const uint16_t* tail;    // mode 0: episode length at episode ends, else 0
  const int64_t* offsets;  // optional row starts (n_rows + 1), else row * ld
  int64_t n_rows, T, ld;
  int lpr;                 // lanes per row
  int mode;                // 0: tail flags, 1: row == one episode
  int P;                   // env_spec.max_episode_length (padding length)
  double gamma, c;         // fp32-rounded discount, discount * gae_lambda
  double gamma_ret;        // discount as the float64 lfilter sees it
  double v0;               // V(zero observation): content of padded baselines
  double bonus_const;      // constant reward bonus, padded cells included
  float* adv;
  float* ret;
};

// Carry entering an episode of length L from its zero-padded tail (Q2).
__device__ double padded_tail_carry(const ScanParams& p, int L, double* boot) {
  const int m = p.P - L;
  if (m <= 0) {
    *boot = 0.0;
    return 0.0;
  }
  *boot = p.v0;
  const double pad_delta = p.bonus_const + (p.gamma - 1.0) * p.v0;
  double cm1 = 1.0, base = p.c;  // c^(m-1) by squaring: cheap in registers
  for (int e = m - 1; e > 0; e >>= 1) {
    if (e & 1) cm1 *= base;
    base *= base;
  }
  const double geo = (p.c == 1.0) ? (double)(m - 1) : (1.0 - cm1) / (1.0 - p.c);
  return pad_delta * geo + cm1 * (p.bonus_const - p.v0);
}

template <bool VEC>
__global__ __launch_bounds__(256, 4) void gae_scan_kernel(ScanParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lpr = p.lpr;
  const int rows_per_wave = 64 / lpr;
  const int sub = lane & (lpr - 1);
  const int64_t row = wave * rows_per_wave + lane / lpr;
  const bool row_ok = row < p.n_rows;

  int64_t start = 0, len = 0;
  if (row_ok) {
    if (p.offsets) {
      start = p.offsets[row];
      len = p.offsets[row + 1] - start;
    } else {
      start = row * p.ld;
      len = p.T;
    }
  }
  // 16-B accesses need the row to start on a 4-float boundary (always true for
  // the padded layout, true for packed batches of fixed-length episodes).
  const bool vec_row = (start & 3) == 0;
  const int C = 4 * lpr;
  int nchunks = (int)((len + C - 1) / C);
  int maxchunks = nchunks;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    maxchunks = max(maxchunks, __shfl_xor(maxchunks, o, 64));

  double carry_a = 0.0, carry_g = 0.0, vnext_chunk = 0.0;

  for (int k = maxchunks - 1; k >= 0; --k) {
    const bool active = k < nchunks;
    const int64_t i0 = (int64_t)k * C + 4 * sub;  // first step of this lane
    float r[4] = {0.f, 0.f, 0.f, 0.f}, v[4] = {0.f, 0.f, 0.f, 0.f};
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    int tl[4] = {0, 0, 0, 0};
    const bool full = active && (i0 + 3 < len);
    if (VEC && vec_row && full) {
      const float4 r4 = *reinterpret_cast<const float4*>(p.rew + start + i0);
      const float4 v4 = *reinterpret_cast<const float4*>(p.val + start + i0);
      r[0] = r4.x; r[1] = r4.y; r[2] = r4.z; r[3] = r4.w;
      v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
      if (p.bonus) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bonus + start + i0);
        b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
      }
      if (p.mode == 0) {
        const ushort4 t4 = *reinterpret_cast<const ushort4*>(p.tail + start + i0);
        tl[0] = t4.x; tl[1] = t4.y; tl[2] = t4.z; tl[3] = t4.w;
      }
    } else if (active) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i0 + j < len) {
          r[j] = p.rew[start + i0 + j];
          v[j] = p.val[start + i0 + j];
          if (p.bonus) b[j] = p.bonus[start + i0 + j];
          if (p.mode == 0) tl[j] = p.tail[start + i0 + j];
        }
      }
    }
    if (p.mode == 1 && active) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i0 + j == len - 1) tl[j] = (int)len;
    }

    // V of the step to the right of this lane's last step.
    double vright = shfl_down_d((double)v[0], 1, lpr);
    if (sub == lpr - 1) vright = vnext_chunk;

    // Per-step affine maps  y_t = x_t + d_t * y_{t+1}.
    // Per-step affine maps  y_t = x_t + d_t * y_{t+1}; d_t is one of three
    // values, so only the offsets are kept in (fp64) registers.
    double xa[4];
    int kind[4];  // 0: beyond the row (identity), 1: episode end, 2: ordinary
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const bool valid = active && (i0 + j < len);
      const double vn = (j == 3) ? vright : (double)v[j + 1];
      const double rb = (double)r[j] + p.bonus_const + (double)b[j];
      if (!valid) {
        xa[j] = 0.0;
        kind[j] = 0;
      } else if (tl[j] > 0) {
        double boot;
        const double c0 = padded_tail_carry(p, tl[j], &boot);
        xa[j] = rb + p.gamma * boot - (double)v[j] + p.c * c0;
        kind[j] = 1;
      } else {
        xa[j] = rb + p.gamma * vn - (double)v[j];
        kind[j] = 2;
      }
    }
    // Compose the lane's four maps (right to left).
    Pair A = {1.0, 0.0}, G = {1.0, 0.0};
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const double da = kind[j] == 2 ? p.c : (kind[j] == 1 ? 0.0 : 1.0);
      const double dg = kind[j] == 2 ? p.gamma_ret : (kind[j] == 1 ? 0.0 : 1.0);
      A.x = xa[j] + da * A.x;
      A.d = da * A.d;
      G.x = (double)r[j] + dg * G.x;
      G.d = dg * G.d;
    }
    // Inclusive suffix scan over the row group.
    for (int o = 1; o < lpr; o <<= 1) {
      const double ad = shfl_down_d(A.d, o, lpr), ax = shfl_down_d(A.x, o, lpr);
      const double gd = shfl_down_d(G.d, o, lpr), gx = shfl_down_d(G.x, o, lpr);
      if (sub + o < lpr) {
        A.x = A.x + A.d * ax;
        A.d = A.d * ad;
        G.x = G.x + G.d * gx;
        G.d = G.d * gd;
      }
    }
    // Carry entering this lane from its right neighbour's suffix.
    double nad = shfl_down_d(A.d, 1, lpr), nax = shfl_down_d(A.x, 1, lpr);
    double ngd = shfl_down_d(G.d, 1, lpr), ngx = shfl_down_d(G.x, 1, lpr);
    double ya = carry_a, yg = carry_g;
    if (sub + 1 < lpr) {
      ya = nax + nad * carry_a;
      yg = ngx + ngd * carry_g;
    }
    float oa[4], og[4];
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const double da = kind[j] == 2 ? p.c : (kind[j] == 1 ? 0.0 : 1.0);
      const double dg = kind[j] == 2 ? p.gamma_ret : (kind[j] == 1 ? 0.0 : 1.0);
      ya = xa[j] + da * ya;
      yg = (double)r[j] + dg * yg;
      oa[j] = (float)ya;
      og[j] = (float)yg;
    }
    if (VEC && vec_row && full) {
      *reinterpret_cast<float4*>(p.adv + start + i0) =
          make_float4(oa[0], oa[1], oa[2], oa[3]);
      *reinterpret_cast<float4*>(p.ret + start + i0) =
          make_float4(og[0], og[1], og[2], og[3]);
    } else if (active) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i0 + j < len) {
          p.adv[start + i0 + j] = oa[j];
          p.ret[start + i0 + j] = og[j];
        }
      }
    }
    // Group-uniform carries for the chunk to the left.
    const double ca = __shfl(ya, 0, lpr), cg = __shfl(yg, 0, lpr);
    const double cv = __shfl((double)v[0], 0, lpr);
    if (active) {
      carry_a = ca;
      carry_g = cg;
      vnext_chunk = cv;
    }
  }
}

// Fast path for rows that are whole episodes of at most 256 steps (mode 1, no
// per-step bonus): fixed-horizon batches, padded rows shorter than P, and packed
// ragged batches.  A row's only special step is its last one, and what the
// zero-padded tail hands it -- bootstrap value V(0-obs) and the closed-form carry
// C0(P - L); nothing when L == P -- is a constant added to that
// step's offset.  After that every step decays by the same constant, so the
// suffix scan carries only the offsets: the decay a lane applies at scan step s is
// c^(4 * 2^s), precomputed on the host by the same repeated squaring the general
// kernel does in registers, and the per-step kind selection disappears: ~5x
// fewer instructions than the general kernel, which is issue bound (not HBM
// bound) at 4096 x 256.  Rows may start at any float (packed batches): the four
// steps of a lane are then loaded and stored one by one, still coalesced across
// the lanes.
struct RowsScanParams {
  const float* rew;
  const float* val;
  const int64_t* offsets;  // n_rows + 1 row starts, or null: row * ld, length T
  int64_t n_rows, T, ld;
  int lpr, P;
  double gamma, c, gamma_ret, bonus_const, v0;
  double cpow[6], gpow[6];
  float* adv;
  float* ret;
};

// NQ = quads (4 steps) per lane: 1 (up to 256 steps per 64-lane row group) or 2
// (8 steps per lane: half the lanes, waves and shuffle steps per row)
template <int NQ>
__global__ __launch_bounds__(256) void gae_scan_rows_kernel(RowsScanParams p) {
  constexpr int NS = 4 * NQ;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lpr = p.lpr;
  const int sub = lane & (lpr - 1);
  const int64_t row = wave * (64 / lpr) + lane / lpr;
  const bool row_ok = row < p.n_rows;
  int64_t start = 0;
  int len = 0;
  if (row_ok) {
    if (p.offsets) {
      start = p.offsets[row];
      len = (int)(p.offsets[row + 1] - start);
    } else {
      start = row * p.ld;
      len = (int)p.T;
    }
  }
  const int i0 = NS * sub;
  const int nv = max(0, min(NS, len - i0));  // valid steps of this lane
  float rf[NS], vf[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) { rf[j] = 0.f; vf[j] = 0.f; }
  const bool vec = nv == NS && ((start + i0) & 3) == 0;
  if (vec) {
    float4 r4[NQ], v4[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      r4[q] = *reinterpret_cast<const float4*>(p.rew + start + i0 + 4 * q);
      v4[q] = *reinterpret_cast<const float4*>(p.val + start + i0 + 4 * q);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      rf[4 * q + 0] = r4[q].x; rf[4 * q + 1] = r4[q].y;
      rf[4 * q + 2] = r4[q].z; rf[4 * q + 3] = r4[q].w;
      vf[4 * q + 0] = v4[q].x; vf[4 * q + 1] = v4[q].y;
      vf[4 * q + 2] = v4[q].z; vf[4 * q + 3] = v4[q].w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (j < nv) {
        rf[j] = p.rew[start + i0 + j];
        vf[j] = p.val[start + i0 + j];
      }
    }
  }
  float vr = __shfl_down(vf[0], 1, lpr);
  if (sub == lpr - 1) vr = 0.f;
  // the row's last step: what the padded tail hands it (only that lane computes)
  double end_add = 0.0;
  const bool has_end = nv > 0 && i0 + nv == len;
  if (has_end) {
    const int m = p.P - len;
    if (m > 0) {
      const double pad_delta = p.bonus_const + (p.gamma - 1.0) * p.v0;
      double cm1 = 1.0, base = p.c;
      for (int e = m - 1; e > 0; e >>= 1) {
        if (e & 1) cm1 *= base;
        base *= base;
      }
      const double geo =
          (p.c == 1.0) ? (double)(m - 1) : (1.0 - cm1) / (1.0 - p.c);
      const double c0 = pad_delta * geo + cm1 * (p.bonus_const - p.v0);
      end_add = p.gamma * p.v0 + p.c * c0;
    }
  }
  double r[NS], xa[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    r[j] = (double)rf[j];
    const bool valid = j < nv;
    const bool last = has_end && j == nv - 1;
    const double vn =
        last ? 0.0 : (double)(j == NS - 1 ? vr : vf[j < NS - 1 ? j + 1 : NS - 1]);
    xa[j] = valid ? (r[j] + p.bonus_const) + p.gamma * vn - (double)vf[j] +
                        (last ? end_add : 0.0)
                  : 0.0;
    if (!valid) r[j] = 0.0;
  }
  double ax = xa[NS - 1], gx = r[NS - 1];
#pragma unroll
  for (int j = NS - 2; j >= 0; --j) {
    ax = xa[j] + p.c * ax;
    gx = r[j] + p.gamma_ret * gx;
  }
  // steps right of the row's end are zero maps with zero offsets: a lane that
  // holds the end must not pass anything on from them (they are zero already)
  int s = 0;
  for (int o = 1; o < lpr; o <<= 1, ++s) {
    const double nax = shfl_down_d(ax, o, lpr), ngx = shfl_down_d(gx, o, lpr);
    if (sub + o < lpr) {
      ax = ax + p.cpow[s] * nax;
      gx = gx + p.gpow[s] * ngx;
    }
  }
  double ya = shfl_down_d(ax, 1, lpr), yg = shfl_down_d(gx, 1, lpr);
  if (sub + 1 >= lpr) {
    ya = 0.0;
    yg = 0.0;
  }
  float oa[NS], og[NS];
#pragma unroll
  for (int j = NS - 1; j >= 0; --j) {
    ya = xa[j] + p.c * ya;
    yg = r[j] + p.gamma_ret * yg;
    oa[j] = (float)ya;
    og[j] = (float)yg;
  }
  if (vec) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      *reinterpret_cast<float4*>(p.adv + start + i0 + 4 * q) =
          make_float4(oa[4 * q], oa[4 * q + 1], oa[4 * q + 2], oa[4 * q + 3]);
      *reinterpret_cast<float4*>(p.ret + start + i0 + 4 * q) =
          make_float4(og[4 * q], og[4 * q + 1], og[4 * q + 2], og[4 * q + 3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (j < nv) {
        p.adv[start + i0 + j] = oa[j];
        p.ret[start + i0 + j] = og[j];
      }
    }
  }
}

}  // namespace

static int g_fixed_fast_path = 1;
// steps per lane of the whole-episode fast path: 4 or 8 (A/B runs)
static int g_rows_steps_per_lane = 4;
extern "C" int ga_set_gae_rows_steps_per_lane(int steps) {
  g_rows_steps_per_lane = steps == 8 ? 8 : 4;
  return 0;
}
extern "C" int ga_set_gae_fixed_fast_path(int on) {
  g_fixed_fast_path = on != 0;
  return 0;
}

// See include/garage_amd.h for the contract.
extern "C" int ga_gae_scan_f32(const float* rewards, const float* values,
                               const float* bonus, const uint16_t* tail,
                               const int64_t* offsets, int64_t n_rows, int64_t T,
                               int64_t ld, int64_t max_len, int mode,
                               int max_episode_length, double discount,
                               double gae_lambda, float v0, float bonus_const,
                               float* adv, float* ret, hipStream_t stream) {
  GA_REQUIRE(rewards && values && adv && ret, "ga_gae_scan_f32: null pointer");
  GA_REQUIRE(n_rows >= 0 && T >= 0, "ga_gae_scan_f32: negative size");
  GA_REQUIRE(mode == 0 || mode == 1, "ga_gae_scan_f32: mode must be 0 or 1");
  GA_REQUIRE(mode == 1 || tail != nullptr,
             "ga_gae_scan_f32: mode 0 needs the tail array");
  GA_REQUIRE(max_episode_length >= 1, "ga_gae_scan_f32: max_episode_length < 1");
  if (!offsets) {
    GA_REQUIRE(ld >= T, "ga_gae_scan_f32: ld (%lld) < T (%lld)", (long long)ld,
               (long long)T);
    max_len = T;
  } else {
    GA_REQUIRE(max_len >= 0, "ga_gae_scan_f32: max_len required with offsets");
  }
  const int64_t prof_steps = T;  // packed mode: callers pass the total step count in T
  if (n_rows == 0 || max_len == 0) return GA_OK;

  int lpr = 1;
  while (lpr < 64 && (int64_t)lpr * 4 < max_len) lpr <<= 1;

  ScanParams p;
  p.rew = rewards; p.val = values; p.bonus = bonus; p.tail = tail;
  p.offsets = offsets; p.n_rows = n_rows; p.T = T; p.ld = ld; p.lpr = lpr;
  p.mode = mode; p.P = max_episode_length;
  // gamma * lambda is formed in fp32 like the reference's filter seed
  // (torch.full(..., discount * gae_lambda, dtype=float), torch/_functions.py:75)
  // and the deltas use the fp32-rounded discount (vpg.py:166 -> :80), while the
  // returns come from a float64 lfilter (np/_functions.py:127).
  p.gamma = (double)(float)discount;
  p.gamma_ret = discount;
  p.c = (double)(float)(discount * gae_lambda);
  p.v0 = (double)v0; p.bonus_const = (double)bonus_const;
  p.adv = adv; p.ret = ret;

  const int rows_per_wave = 64 / lpr;
  const int64_t waves = ga_ceil_div(n_rows, rows_per_wave);
  const int64_t blocks = ga_ceil_div(waves, 4);
  GA_REQUIRE(blocks < (1ll << 31), "ga_gae_scan_f32: grid too large");
  // whole-episode rows of at most 256 steps: the constant-decay kernel
  if (g_fixed_fast_path && mode == 1 && !bonus && max_len <= 256) {
    const int nq = (g_rows_steps_per_lane == 8 && max_len > 4) ? 2 : 1;
    int flpr = 1;
    while (flpr < 64 && (int64_t)flpr * 4 * nq < max_len) flpr <<= 1;
    RowsScanParams f;
    f.rew = rewards; f.val = values; f.offsets = offsets; f.n_rows = n_rows; f.T = T;
    f.ld = ld; f.lpr = flpr; f.P = max_episode_length; f.gamma = p.gamma; f.c = p.c;
    f.gamma_ret = p.gamma_ret; f.bonus_const = p.bonus_const; f.v0 = p.v0;
    f.adv = adv; f.ret = ret;
    // decay over one lane's steps, then squared per scan step
    double cd = 1.0, gd = 1.0;
    for (int k = 0; k < 4 * nq; ++k) {
      cd *= p.c;
      gd *= discount;
    }
    for (int k = 0; k < 6; ++k) {
      f.cpow[k] = cd;
      f.gpow[k] = gd;
      cd *= cd;
      gd *= gd;
    }
    const int64_t fwaves = ga_ceil_div(n_rows, 64 / flpr);
    const int64_t fblocks = ga_ceil_div(fwaves, 4);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ga_prof_events(GA_PROF_GAE_SCAN,
                   16.0 * (offsets ? (double)prof_steps : (double)n_rows * (double)T),
                   &e0, &e1);
    if (nq == 2)
      hipExtLaunchKernelGGL(gae_scan_rows_kernel<2>, dim3((unsigned)fblocks), dim3(256),
                            0, stream, e0, e1, 0, f);
    else
      hipExtLaunchKernelGGL(gae_scan_rows_kernel<1>, dim3((unsigned)fblocks), dim3(256),
                            0, stream, e0, e1, 0, f);
    GA_CHECK_LAUNCH("ga_gae_scan_f32");
    return GA_OK;
  }
  const bool vec = (offsets || ld % 4 == 0) && ga_aligned16(rewards) &&
                   ga_aligned16(values) && ga_aligned16(adv) && ga_aligned16(ret) &&
                   (!bonus || ga_aligned16(bonus)) &&
                   (!tail || (reinterpret_cast<uintptr_t>(tail) & 7u) == 0);
  // algorithmic bytes: 16 per step (r, V in; A, G out)
  const double steps = offsets ? 0.0 : (double)n_rows * (double)T;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ga_prof_events(GA_PROF_GAE_SCAN, 16.0 * (offsets ? (double)prof_steps : steps),
                 &e0, &e1);
  if (vec)
    hipExtLaunchKernelGGL(gae_scan_kernel<true>, dim3((unsigned)blocks), dim3(256),
                          0, stream, e0, e1, 0, p);
  else
    hipExtLaunchKernelGGL(gae_scan_kernel<false>, dim3((unsigned)blocks), dim3(256),
                          0, stream, e0, e1, 0, p);
  GA_CHECK_LAUNCH("ga_gae_scan_f32");
  return GA_OK;
}
