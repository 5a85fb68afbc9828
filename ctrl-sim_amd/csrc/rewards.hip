// Real-time reward ledger of the Decision-Transformer baseline on the device (gfx950): the RTG the policy is conditioned on at
// step t, from the dense reward of step t-1, for every vehicle of every scenario — one wave per vehicle, float64.
//
// Reference: cfgs/policy/dt.yaml (real_time_rewards, max_return) drives
//   evaluators/policy_evaluator.py:122-147   RTG_0 = (10, 90, 90) [max_return] / (0, -10, -10) [min_return, evaluated vehicles];
//                                            RTG_t = RTG_{t-1} - dense_reward_{t-1}
//   evaluators/evaluator.py:106-140          compute_dense_reward: positions / existence of the CURRENT step, but the reward row it
//                                            reads is `all_rewards[i, 0]` — the goal / collision flags of STEP 0 (kept as written)
//   datasets/rl_waymo/dataset.py:187-275     compute_dist_to_nearest_road_edge_rewards, compute_dist_to_nearest_vehicle_rewards
//                                            (normalize=False), compute_rewards (goal, vehicle, road-edge components)
//   utils/data.py:152-290                    compute_distance_to_road_edge / signed distance to polylines
//   policies/autoregressive_policy.py:73-78  clip + scale of the three RTG components to [0, 1] before the model sees them
//   utils/sim.py:83-141                      compute_reward at step 0 (goal-reached flag, shaped goal term, collision flags)
// The road-edge component only uses |signed distance to the nearest road-edge polyline| (dataset.py:268 takes np.abs), and the
// nearest polyline is the one of smallest |distance| (utils/data.py:178-181), so it is the minimum point-segment distance over all
// road-edge segments — the segment table the simulator already holds (`edges`).  The sign logic of utils/data.py:252-287 drops out
// except when a point is exactly collinear with its nearest segment (sign 0 -> distance 0): not reproduced (measure zero).
// Arithmetic follows the NumPy expressions operation by operation (no FMA contraction), so the ledger agrees with the host
// implementation (ctrlsim_amd/rewards.py, pinned by tests/golden/dense_reward.npz) to the last bits that the reductions allow.
#include "common.h"
#include "../../include/ctrlsim.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// ledger row [10]: rtg[3] (raw, of the step just written), dense[3] (of that step), rew0[4] = reached, shaped, coll_veh, coll_edge at step 0
__global__ __launch_bounds__(64) void dt_ledger_kernel(int N, int E, int t, int T1, int Tmax, const float* __restrict__ hist_states,
                                                       const unsigned char* __restrict__ coll, const double* __restrict__ goals,
                                                       const float* __restrict__ edges, const double* __restrict__ init_rtg,
                                                       ctrlsim_dt_reward_cfg c, double* __restrict__ ledger,
                                                       double* __restrict__ rtg_raw, int* __restrict__ hist_rtg) {
  const int s = blockIdx.x / N, v = blockIdx.x % N, lane = threadIdx.x;
  const float* row = hist_states + (((size_t)s * N + v) * T1 + t) * 8;
  const double x = row[0], y = row[1], ex = row[7];
  // ---- |distance| to the nearest road-edge segment (utils/data.py:236-247: rel_t = nan_to_num(s2p.s2e / s2e.s2e), clipped)
  double dmin2 = __builtin_inf();
  const float* eg = edges + (size_t)s * E * 4;
  for (int e = lane; e < E; e += 64) {
    const double x0 = eg[e * 4], y0 = eg[e * 4 + 1], x1 = eg[e * 4 + 2], y1 = eg[e * 4 + 3];
    if (x0 > 1e29) continue;                                    // padding rows of the segment table
    const double px = x - x0, py = y - y0, sx = x1 - x0, sy = y1 - y0;
    const double den = sx * sx + sy * sy, num = px * sx + py * sy;
    double r = den != 0.0 ? num / den : 0.0;                    // zero-length segment: 0 / 0 -> nan_to_num -> 0
    r = clipd(r, 0.0, 1.0);
    const double qx = px - sx * r, qy = py - sy * r;
    dmin2 = fmin(dmin2, qx * qx + qy * qy);
  }
  const double edge_dist = sqrt(wave_min(dmin2));               // inf when the scenario has no road edge
  // ---- distance to the nearest other existing vehicle (dataset.py:202-237, normalize=False): 0 when alone / not existing
  double n2 = __builtin_inf();
  for (int u = lane; u < N; u += 64) {
    if (u == v) continue;
    const float* ru = hist_states + (((size_t)s * N + u) * T1 + t) * 8;
    if (ru[7] == 0.f) continue;
    const double dx = x - (double)ru[0], dy = y - (double)ru[1];
    n2 = fmin(n2, dx * dx + dy * dy);
  }
  n2 = wave_min(n2);
  if (lane != 0) return;
  double nearest = (ex != 0.0 && n2 < __builtin_inf()) ? sqrt(n2) * ex : 0.0;
  nearest = nearest * ex;
  double* L = ledger + ((size_t)s * N + v) * 10;
  const unsigned char* cl = coll + (((size_t)s * N + v) * T1 + t) * 2;
  double rtg[3];
  if (t == 0) {
    // step-0 reward row (utils/sim.py:83-141 with no earlier step: the goal was not reached before, the shaped goal term is
    // 1 - dist/dist = 0 unless the vehicle starts on its goal, where the normaliser falls back to 1)
    const double* g = goals + ((size_t)s * N + v) * 5;
    const double gx = g[0] - x, gy = g[1] - y, dist0 = sqrt(gx * gx + gy * gy);
    L[6] = dist0 < c.pos_tol ? 1.0 : 0.0;
    L[7] = dist0 == 0.0 ? c.shaped_unit : c.shaped_unit * (1.0 - dist0 / dist0);
    L[8] = cl[0] ? 1.0 : 0.0;
    L[9] = cl[1] ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) rtg[k] = init_rtg ? init_rtg[((size_t)s * N + v) * 3 + k] : (k == 0 ? 10.0 : 90.0);
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) rtg[k] = L[k] - L[3 + k];
  }
  // ---- dense reward of this step (dataset.py:239-275 on the step-0 row scaled by the current existence, evaluator.py:131-138)
  const double r0 = L[6] * ex, r3 = L[7] * ex, r6 = L[8] * ex, r7 = L[9] * ex;
  const double veh = clipd(nearest, 0.0, c.max_veh_dist) / c.max_veh_dist;
  double goal = r0 * c.goal_mult;
  if (!c.remove_shaped_goal) goal = goal + (clipd(r3, c.shaped_min, c.shaped_max) - c.shaped_max) / c.shaped_max;
  const double vv = c.remove_shaped_veh ? -r6 * c.veh_mult : veh - r6 * c.veh_mult;
  // evaluator.py:124-125 divides the distance by the scaling factor (and masks it), dataset.py:268 multiplies |.| back
  const double ed = ex != 0.0 ? (edge_dist / c.edge_scale) * ex : 0.0;
  const double ve = c.remove_shaped_edge ? -r7 * c.edge_mult : clipd(ed * c.edge_scale, 0.0, 5.0) / 5.0 - r7 * c.edge_mult;
  L[3] = goal * ex; L[4] = vv * ex; L[5] = ve * ex;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    L[k] = rtg[k];
    if (rtg_raw) rtg_raw[(((size_t)s * N + v) * Tmax + t) * 3 + k] = rtg[k];
    const float nrm = (float)((clipd(rtg[k], c.rtg_lo[k], c.rtg_hi[k]) - c.rtg_lo[k]) / (c.rtg_hi[k] - c.rtg_lo[k]));
    hist_rtg[(((size_t)s * N + v) * Tmax + t) * 3 + k] = __float_as_int(nrm);
  }
}

}  // namespace

// hist_rtg[s, v, t, :] <- float bits of the clip-normalised RTG of step t; ledger [S,N,10] float64 carries RTG / dense reward /
// step-0 flags from call to call (t = 0 initialises it); rtg_raw [S,N,Tmax,3] float64 (nullable) records the raw RTGs.
extern "C" int ctrlsim_dt_ledger_step(int S, int N, int E, int t, int T1, int Tmax, const float* hist_states, const unsigned char* coll,
                                      const double* goals, const float* edges, const double* init_rtg,
                                      const ctrlsim_dt_reward_cfg* cfg, double* ledger, double* rtg_raw, int* hist_rtg,
                                      hipStream_t st) {
  if (S < 0 || N < 1 || N > 64 || E < 0 || t < 0 || t >= Tmax || T1 <= t || !hist_states || !coll || !goals || (E && !edges) || !cfg ||
      !ledger || !hist_rtg)
    return CTRLSIM_EINVAL;
  if (S == 0) return CTRLSIM_OK;
  prof_before(PROF_CTX, st);
  hipLaunchKernelGGL(dt_ledger_kernel, dim3(S * N), dim3(64), 0, st, N, E, t, T1, Tmax, hist_states, coll, goals, edges, init_rtg, *cfg,
                     ledger, rtg_raw, hist_rtg);
  prof_after(PROF_CTX, 0.0, st, (double)S * N * (16.0 * E + 32.0 * N + 200.0));
  return ctrlsim_launch_status();
}
