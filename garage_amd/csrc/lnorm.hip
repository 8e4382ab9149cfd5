// Layer normalisation in front of a hidden linear layer (gfx950).
//
// The reference's MLP blocks are  LayerNorm(prev) -> Linear -> nonlinearity  when
// layer_normalization=True (torch/modules/multi_headed_mlp_module.py:77-92,
// nn.LayerNorm defaults: eps = 1e-5, elementwise affine, biased variance over the
// feature dimension).  Rows are at most a few hundred floats wide: one wave per
// row, the row in registers.
//
//   ln_fwd_kernel   y = (x - mean) * rstd * gamma + beta ; (mean, rstd) kept per row
//   ln_bwd_kernel   from dy = d(loss)/dy:
//                     dgamma += dy * xhat,  dbeta += dy       (per row block -> slab)
//                     dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//                   and, when the rows are the outputs h of a previous layer,
//                   dz = dx * slope(h) for that layer's backward pass (in place over
//                   dy).
// Sums over the rows of a block are taken in a fixed order (bitwise reproducible).
#include "common.h"

#include "gemm_core.h"

namespace {

constexpr int LN_MAXW = 1024;            // widest normalised row
constexpr int LN_PER_LANE = LN_MAXW / 64;
constexpr float LN_EPS = 1e-5f;

struct LnFwdParams {
  const float* X; int64_t ldx; const int32_t* idx;
  int64_t M; int D;
  const float* gamma; const float* beta;
  float* Y; int64_t ldy;
  float* stats;  // [M][2]: mean, rstd
};

__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= p.M) return;
  const int64_t src = p.idx ? (int64_t)p.idx[row] : row;
  const float* x = p.X + src * p.ldx;
  float v[LN_PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_PER_LANE; ++k) {
    const int j = lane + 64 * k;
    v[k] = j < p.D ? x[j] : 0.f;
    s += v[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)p.D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_PER_LANE; ++k) {
    const int j = lane + 64 * k;
    const float d = j < p.D ? v[k] - mean : 0.f;
    q += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.f / sqrtf(q / (float)p.D + LN_EPS);
  float* y = p.Y + row * p.ldy;
#pragma unroll
  for (int k = 0; k < LN_PER_LANE; ++k) {
    const int j = lane + 64 * k;
    if (j < p.D) y[j] = (v[k] - mean) * rstd * p.gamma[j] + p.beta[j];
    else if (j < (int)p.ldy) y[j] = 0.f;  // padding columns stay zero
  }
  if (lane == 0) {
    p.stats[2 * row] = mean;
    p.stats[2 * row + 1] = rstd;
  }
}

struct LnBwdParams {
  float* dY; int64_t ldd;      // in: d(loss)/dy ; out (want_dx): dz of the previous layer
  const float* X; int64_t ldx; const int32_t* idx;  // the rows that were normalised
  const float* stats;
  int64_t M; int D;
  const float* gamma;
  int want_dx;                 // 0: first layer (the observations take no gradient)
  int hact;                    // slope code of the activation that produced X (network
                               // code: 0 tanh, 1 relu, 2 none)
  int rows_per_split;
  float* dgamma; float* dbeta; // slab 0 addresses; + split * split_stride
  int64_t split_stride;
};

// one workgroup (4 waves) per split of rows; wave w takes rows w, w + 4, ...
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdParams p) {
  __shared__ float red[2][4][LN_MAXW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_split;
  const int64_t r1 = min(p.M, r0 + p.rows_per_split);
  float gsum[LN_PER_LANE], bsum[LN_PER_LANE], gam[LN_PER_LANE];
#pragma unroll
  for (int k = 0; k < LN_PER_LANE; ++k) {
    const int j = lane + 64 * k;
    gsum[k] = 0.f; bsum[k] = 0.f;
    gam[k] = j < p.D ? p.gamma[j] : 0.f;
  }
  for (int64_t row = r0 + wave; row < r1; row += 4) {
    const int64_t src = p.idx ? (int64_t)p.idx[row] : row;
    const float* x = p.X + src * p.ldx;
    float* dy = p.dY + row * p.ldd;
    const float mean = p.stats[2 * row], rstd = p.stats[2 * row + 1];
    float xv[LN_PER_LANE], xh[LN_PER_LANE], g[LN_PER_LANE];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_PER_LANE; ++k) {
      const int j = lane + 64 * k;
      const bool ok = j < p.D;
      xv[k] = ok ? x[j] : 0.f;
      const float d = ok ? dy[j] : 0.f;
      xh[k] = ok ? (xv[k] - mean) * rstd : 0.f;
      gsum[k] += d * xh[k];
      bsum[k] += d;
      g[k] = d * gam[k];
      c1 += g[k];
      c2 += g[k] * xh[k];
    }
    if (p.want_dx) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        c1 += __shfl_xor(c1, o, 64);
        c2 += __shfl_xor(c2, o, 64);
      }
      c1 /= (float)p.D;
      c2 /= (float)p.D;
#pragma unroll
      for (int k = 0; k < LN_PER_LANE; ++k) {
        const int j = lane + 64 * k;
        if (j < p.D)
          dy[j] = rstd * (g[k] - c1 - xh[k] * c2) * act_slope(xv[k], p.hact);
      }
    }
  }
  // the four waves' column sums meet in LDS, added in wave order
#pragma unroll
  for (int k = 0; k < LN_PER_LANE; ++k) {
    const int j = lane + 64 * k;
    red[0][wave][j] = gsum[k];
    red[1][wave][j] = bsum[k];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < p.D; j += 256) {
    const float a = ((red[0][0][j] + red[0][1][j]) + red[0][2][j]) + red[0][3][j];
    const float b = ((red[1][0][j] + red[1][1][j]) + red[1][2][j]) + red[1][3][j];
    p.dgamma[(int64_t)blockIdx.x * p.split_stride + j] = a;
    p.dbeta[(int64_t)blockIdx.x * p.split_stride + j] = b;
  }
}

// Tangent of y = LN(x) for the tangents (tx, tgamma, tbeta):
//   ty = gamma rstd (tx - mean(tx) - xhat mean(tx xhat)) + tgamma xhat + tbeta
// (tx == null: the rows are inputs of the network and carry no tangent)
struct LnJvpParams {
  const float* tX; int64_t ldt;
  const float* X; int64_t ldx; const int32_t* idx;
  const float* stats;
  int64_t M; int D;
  const float* gamma; const float* tgamma; const float* tbeta;
  float* tY; int64_t ldy;
};

__global__ __launch_bounds__(256) void ln_jvp_kernel(LnJvpParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= p.M) return;
  const int64_t src = p.idx ? (int64_t)p.idx[row] : row;
  const float* x = p.X + src * p.ldx;
  const float mean = p.stats[2 * row], rstd = p.stats[2 * row + 1];
  float xh[LN_PER_LANE], tx[LN_PER_LANE];
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int k = 0; k < LN_PER_LANE; ++k) {
    const int j = lane + 64 * k;
    const bool ok = j < p.D;
    xh[k] = ok ? (x[j] - mean) * rstd : 0.f;
    tx[k] = (ok && p.tX) ? p.tX[row * p.ldt + j] : 0.f;
    c1 += tx[k];
    c2 += tx[k] * xh[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c1 += __shfl_xor(c1, o, 64);
    c2 += __shfl_xor(c2, o, 64);
  }
  c1 /= (float)p.D;
  c2 /= (float)p.D;
  float* ty = p.tY + row * p.ldy;
#pragma unroll
  for (int k = 0; k < LN_PER_LANE; ++k) {
    const int j = lane + 64 * k;
    if (j < p.D)
      ty[j] = p.gamma[j] * rstd * (tx[k] - c1 - xh[k] * c2) + p.tgamma[j] * xh[k] +
              p.tbeta[j];
    else if (j < (int)p.ldy)
      ty[j] = 0.f;
  }
}

}  // namespace

int ga_ln_jvp(const float* tX, int64_t ldt, const float* X, int64_t ldx, const int32_t* idx,
              const float* stats, int64_t M, int D, const float* gamma,
              const float* tgamma, const float* tbeta, float* tY, int64_t ldy,
              hipStream_t stream) {
  GA_REQUIRE(X && stats && gamma && tgamma && tbeta && tY && M >= 1 && D >= 1 &&
                 D <= LN_MAXW && ldy >= D && ldy <= LN_MAXW,
             "layer normalisation tangent: bad arguments");
  LnJvpParams p;
  p.tX = tX; p.ldt = ldt; p.X = X; p.ldx = ldx; p.idx = idx; p.stats = stats; p.M = M;
  p.D = D; p.gamma = gamma; p.tgamma = tgamma; p.tbeta = tbeta; p.tY = tY; p.ldy = ldy;
  hipLaunchKernelGGL(ln_jvp_kernel, dim3((unsigned)ga_ceil_div(M, 4)), dim3(256), 0, stream,
                     p);
  GA_CHECK_LAUNCH("ln_jvp");
  return GA_OK;
}

// Internal entry points (gemm.hip calls them from the per-layer forward / backward).
int ga_ln_forward(const float* X, int64_t ldx, const int32_t* idx, int64_t M, int D,
                  const float* gamma, const float* beta, float* Y, int64_t ldy,
                  float* stats, hipStream_t stream) {
  GA_REQUIRE(X && gamma && beta && Y && stats && M >= 1 && D >= 1 && D <= LN_MAXW &&
                 ldy >= D && ldy <= LN_MAXW,
             "layer normalisation: unsupported row width %d", D);
  LnFwdParams p;
  p.X = X; p.ldx = ldx; p.idx = idx; p.M = M; p.D = D; p.gamma = gamma; p.beta = beta;
  p.Y = Y; p.ldy = ldy; p.stats = stats;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)ga_ceil_div(M, 4)), dim3(256), 0, stream,
                     p);
  GA_CHECK_LAUNCH("ln_fwd");
  return GA_OK;
}

// n_splits blocks, block s reduces rows [s * rows_per_split, ...) into slab s
int ga_ln_backward(float* dY, int64_t ldd, const float* X, int64_t ldx, const int32_t* idx,
                   const float* stats, int64_t M, int D, const float* gamma, int want_dx,
                   int hact, int rows_per_split, int n_splits, float* dgamma, float* dbeta,
                   int64_t split_stride, hipStream_t stream) {
  GA_REQUIRE(dY && X && stats && gamma && dgamma && dbeta && M >= 1 && D >= 1 &&
                 D <= LN_MAXW && rows_per_split >= 1 && n_splits >= 1,
             "layer normalisation backward: bad arguments");
  LnBwdParams p;
  p.dY = dY; p.ldd = ldd; p.X = X; p.ldx = ldx; p.idx = idx; p.stats = stats; p.M = M;
  p.D = D; p.gamma = gamma; p.want_dx = want_dx; p.hact = hact;
  p.rows_per_split = rows_per_split; p.dgamma = dgamma; p.dbeta = dbeta;
  p.split_stride = split_stride;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)n_splits), dim3(256), 0, stream, p);
  GA_CHECK_LAUNCH("ln_bwd");
  return GA_OK;
}
