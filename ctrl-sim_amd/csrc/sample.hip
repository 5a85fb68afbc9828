// Token sampling for the closed-loop rollout: RTG bins (3 heads of 350) and action tokens (1000) (gfx950).
//
// Reference: policies/policy.py:108-142 (process_predicted_rtg: logits reshaped (350,3) bin-major, + tilt*linspace(0,1),
// softmax, torch.multinomial), policies/autoregressive_policy.py:214-240 (temperature softmax, optional nucleus top-p,
// torch.multinomial), datasets/rl_waymo/dataset.py:340-347 (get_tilt_logits).
// torch.multinomial(p, 1) is the exponential race argmax_i p_i / q_i, q ~ Exp(1) (aten multinomial kernel, verified
// against torch in tests/golden/sampling.npz).  argmax_i p_i/q_i == argmax_i (x_i - log q_i) for p = softmax(x): the
// kernel evaluates that score in float64, so no normaliser is needed and the result is the reference's draw
// whenever the reference's own float rounding does not decide the race (fixtures assert the race margins).
// The noise q is either supplied (parity tests) or generated in-kernel from the counter-based generator shared with
// ctrlsim_amd/weights.py: key = H(seed, scenario, t, agent, head), u = (top24(splitmix64(key + i*C)) + 0.5) / 2^24,
// q = (float)(-log u).
// One wavefront per (scenario, vehicle); lanes stride over the vocabulary.
#include "common.h"

__device__ __forceinline__ uint64_t noise_key(uint64_t seed, uint64_t scenario, uint64_t t, uint64_t agent,
                                              uint64_t head) {
  uint64_t k = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  k = splitmix64(k ^ scenario);
  k = splitmix64(k ^ t);
  k = splitmix64(k ^ agent);
  k = splitmix64(k ^ head);
  return k;
}

__device__ __forceinline__ float exp_noise(uint64_t key, int i) {
  const uint64_t z = splitmix64(key + (uint64_t)i * 0xD1342543DE82EF95ull);
  const double u = ((double)(z >> 40) + 0.5) / 16777216.0;
  return (float)(-log(u));
}

struct ArgBest {
  double s;
  int i;
};

// Races that no finite score won (NaN logits: e.g. an activation beyond the fp16 range of the split operands, csrc/split.h):
// counted in the process-wide non-finite counter (common.h), the token falls back to a valid id, and ctrlsim_nonfinite_count
// lets the host fail loudly.

__device__ __forceinline__ ArgBest wave_argmax(double s, int i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double s2 = __shfl_xor(s, o, 64);
    const int i2 = __shfl_xor(i, o, 64);
    if (s2 > s || (s2 == s && i2 < i)) { s = s2; i = i2; }    // ties -> lowest index, like argmax
  }
  return {s, i};
}

// rtg_logits [Bctx, A, R*3] — or, with ctx_row0 (compact contexts of different slot counts), logits rows, context c's slot s at
// row ctx_row0[c] + s; own_ctx/own_slot/tilted [S*N]; hist_rtg [S,N,Tmax,3]
__global__ __launch_bounds__(256) void sample_rtg_kernel(const float* __restrict__ rtg_logits, int A, int R,
                                                         const int* __restrict__ own_ctx,
                                                         const int* __restrict__ own_slot,
                                                         const int* __restrict__ ctx_row0,
                                                         const unsigned char* __restrict__ tilted, double tilt_goal,
                                                         double tilt_veh, double tilt_road,
                                                         const double* __restrict__ tilt_scn,  // [S,3] or null (uniform)
                                                         const float* __restrict__ noise,      // [S*N, 3, R] or null
                                                         uint64_t seed, const int64_t* __restrict__ scenario_id, int t,
                                                         int* __restrict__ hist_rtg, int N, int Tmax, int SN,
                                                         int* __restrict__ g_nonfinite) {
  const int sv = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sv >= SN) return;
  const int ctx = own_ctx[sv];
  if (ctx < 0) return;                              // vehicle is in no context this step: row keeps the (0,35,35) default
  const int lane = threadIdx.x & 63;
  const int s = sv / N, v = sv - s * N;
  const float* lg = rtg_logits + ((ctx_row0 ? (size_t)ctx_row0[ctx] : (size_t)ctx * A) + own_slot[sv]) * (size_t)(R * 3);
  const bool tl = tilted[sv] != 0;
  if (tilt_scn) { tilt_goal = tilt_scn[3 * s]; tilt_veh = tilt_scn[3 * s + 1]; tilt_road = tilt_scn[3 * s + 2]; }
  const double tilts[3] = {tl ? tilt_goal : 0.0, tl ? tilt_veh : 0.0, tl ? tilt_road : 0.0};
  const double step = 1.0 / (double)(R - 1);        // np.linspace(0, 1, R)[i] = i * step  (endpoint exact)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const uint64_t key = noise ? 0 : noise_key(seed, (uint64_t)scenario_id[s], (uint64_t)t, (uint64_t)v, (uint64_t)c);
    double best = -__builtin_inf();
    int bi = 0x7fffffff;
    for (int i = lane; i < R; i += 64) {
      const float q = noise ? noise[((size_t)sv * 3 + c) * R + i] : exp_noise(key, i);
      const double lin = (i == R - 1) ? 1.0 : (double)i * step;
      const double sc = ((double)lg[i * 3 + c] + tilts[c] * lin) - log((double)q);
      if (sc > best) { best = sc; bi = i; }
    }
    ArgBest r = wave_argmax(best, bi);
    if (r.i >= R) { r.i = 0; if (lane == 0) atomicAdd(g_nonfinite, 1); }
    if (lane == 0) hist_rtg[((size_t)sv * Tmax + t) * 3 + c] = r.i;
  }
}

// act_logits [Bctx, A, V]; mem_ctx/mem_slot [S*N] (-1: not evaluated this step -> zero action)
__global__ __launch_bounds__(256) void sample_action_kernel(const float* __restrict__ act_logits, int A, int V,
                                                            const int* __restrict__ mem_ctx,
                                                            const int* __restrict__ mem_slot,
                                                            const int* __restrict__ ctx_row0, float temperature,
                                                            double top_p, const float* __restrict__ noise,  // [S*N, V] or null
                                                            uint64_t seed, const int64_t* __restrict__ scenario_id,
                                                            int t, int* __restrict__ hist_tok, int* __restrict__ act_now,
                                                            int N, int Tmax, int SN, int zero_token,
                                                            int* __restrict__ g_nonfinite) {
  extern __shared__ double pbuf[];                  // [4][V] when nucleus sampling is on
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sv = blockIdx.x * 4 + w;
  if (sv >= SN) return;
  const int ctx = mem_ctx[sv];
  if (ctx < 0) {                                    // dead / not evaluated: (0,0) action, autoregressive_policy.py:249-251
    if (lane == 0) { hist_tok[(size_t)sv * Tmax + t] = zero_token; act_now[sv] = -1; }
    return;
  }
  const int s = sv / N, v = sv - s * N;
  const float* lg = act_logits + ((ctx_row0 ? (size_t)ctx_row0[ctx] : (size_t)ctx * A) + mem_slot[sv]) * (size_t)V;
  const uint64_t key = noise ? 0 : noise_key(seed, (uint64_t)scenario_id[s], (uint64_t)t, (uint64_t)v, 3ull);
  double* p = pbuf + (size_t)w * V;
  const bool nucleus = top_p > 0.0;
  if (nucleus) {
    // probabilities (float64 softmax of the float32 scaled logits), then rank-cumulative mass per token
    float mx = -__builtin_inff();
    for (int i = lane; i < V; i += 64) mx = fmaxf(mx, lg[i] / temperature);
    mx = wave_max(mx);
    double z = 0.0;
    for (int i = lane; i < V; i += 64) {
      const double e = exp((double)(lg[i] / temperature) - (double)mx);
      p[i] = e;
      z += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < V; i += 64) p[i] = p[i] / z;
    __builtin_amdgcn_wave_barrier();
  }
  double best = -__builtin_inf();
  int bi = 0x7fffffff;
  for (int i = lane; i < V; i += 64) {
    if (nucleus) {
      const double pi = p[i];
      double before = 0.0;                          // mass of the tokens sorted ahead of i (descending p, index tiebreak)
      for (int j = 0; j < V; ++j) {
        const double pj = p[j];
        if (pj > pi || (pj == pi && j < i)) before += pj;
      }
      if (!(before < top_p)) continue;              // outside the top-p set (autoregressive_policy.py:217-231)
    }
    const float q = noise ? noise[(size_t)sv * V + i] : exp_noise(key, i);
    const double sc = (double)(lg[i] / temperature) - log((double)q);
    if (sc > best) { best = sc; bi = i; }
  }
  ArgBest r = wave_argmax(best, bi);
  if (r.i >= V) { r.i = zero_token; if (lane == 0) atomicAdd(g_nonfinite, 1); }
  if (lane == 0) { hist_tok[(size_t)sv * Tmax + t] = r.i; act_now[sv] = r.i; }
}

int launch_sample_rtg(const float* rtg_logits, int A, int R, const int* own_ctx, const int* own_slot, const int* ctx_row0,
                      const unsigned char* tilted, const double* tilt3, const double* tilt_scn, const float* noise, uint64_t seed,
                      const int64_t* scenario_id, int t, int* hist_rtg, int S, int N, int Tmax, hipStream_t st) {
  const int SN = S * N;
  if (SN <= 0) return CTRLSIM_OK;
  if (t < 0 || t >= Tmax) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(sample_rtg_kernel, dim3((SN + 3) / 4), dim3(256), 0, st, rtg_logits, A, R, own_ctx, own_slot, ctx_row0, tilted,
                     tilt3[0], tilt3[1], tilt3[2], tilt_scn, noise, seed, scenario_id, t, hist_rtg, N, Tmax, SN, ctrlsim_nonfinite_ptr());
  return ctrlsim_launch_status();
}

int launch_sample_action(const float* act_logits, int A, int V, const int* mem_ctx, const int* mem_slot, const int* ctx_row0,
                         float temperature, double top_p, const float* noise, uint64_t seed,
                         const int64_t* scenario_id, int t, int* hist_tok, int* act_now, int S, int N, int Tmax,
                         int zero_token, hipStream_t st) {
  const int SN = S * N;
  if (SN <= 0) return CTRLSIM_OK;
  if (t < 0 || t >= Tmax || temperature <= 0.f) return CTRLSIM_EINVAL;
  const size_t shm = top_p > 0.0 ? (size_t)4 * V * sizeof(double) : 0;
  hipLaunchKernelGGL(sample_action_kernel, dim3((SN + 3) / 4), dim3(256), shm, st, act_logits, A, V, mem_ctx, mem_slot, ctx_row0,
                     temperature, top_p, noise, seed, scenario_id, t, hist_tok, act_now, N, Tmax, SN, zero_token,
                     ctrlsim_nonfinite_ptr());
  return ctrlsim_launch_status();
}

