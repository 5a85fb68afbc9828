// Rollout metrics on the device: the running statistics of the evaluator for S finished rollouts, accumulated into ONE packed
// float64 vector — the payload of the only collective of the multi-GPU rollout (gfx950).  -ffp-contract=off.
//
// Reference (host Python, float64):
//   evaluators/policy_evaluator.py:162-248  update_running_statistics: per evaluated vehicle over the existing steps >= history_steps:
//                                           goal reached (latched reward[0]), any vehicle / road-edge collision, ADE, FDE, samples
//                                           of speed, "angular speed" (= heading / dt as written, :219-220), acceleration (first and
//                                           last sample dropped), nearest-vehicle distance, for the rollout and for the log
//   utils/sim.py:83-141                     compute_reward: position target within tolerance, latched once reached
//   evaluators/evaluator.py:87-103 +
//   datasets/rl_waymo/dataset.py:202-237    nearest-vehicle distance among existing vehicles (0 when alone)
//   evaluators/policy_evaluator.py:251-305  the histograms compute_metrics builds (np.histogram over fixed edges, values clipped)
// Layout of `out` (+=, float64) = ctrlsim_amd/metrics.py MetricAccumulators.pack(): sums then counts of (goal, collision rate,
// off-road rate, ADE, FDE), then the histograms accel_gt[20] accel_sim[20] ang_gt[200] ang_sim[200] lin_gt[200] lin_sim[200]
// nd_gt[200] nd_sim[200].  One workgroup (one wavefront, thread = vehicle) per scenario; the N x N nearest-distance search per
// step reads the other vehicles' rows through the cache.
#include "common.h"

namespace {
struct MetricParams {
  int N, T1, Tmax, hist_steps, n_accel, n_steer, acc_bins;
  double dt, pos_tol, min_accel, max_accel;
};
enum { M_SUMS = 0, M_COUNTS = 5, M_ACC_GT = 10, M_ACC_SIM = 30, M_ANG_GT = 50, M_ANG_SIM = 250, M_LIN_GT = 450, M_LIN_SIM = 650,
       M_ND_GT = 850, M_ND_SIM = 1050, M_TOTAL = 1250 };

// np.histogram over explicit edges e[0..n]: bin i holds e_i <= v < e_{i+1}, the last bin also v == e_n
__device__ __forceinline__ void hist_add(double* h, const double* e, int n, double v) {
  if (!(v >= e[0]) || v > e[n]) return;
  int lo = 0, hi = n;                                   // largest i with e[i] <= v
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (e[mid] <= v) lo = mid; else hi = mid;
  }
  atomicAdd(&h[lo], 1.0);
}
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

__global__ __launch_bounds__(64) void metrics_pack_kernel(MetricParams p, const float* __restrict__ states,   // [S,N,T1,8]
                                                          const unsigned char* __restrict__ coll,             // [S,N,T1,2]
                                                          const int* __restrict__ tok,                        // [S,N,Tmax]
                                                          const double* __restrict__ gt,                      // [S,N,T1,5] x,y,heading,speed,exist
                                                          const double* __restrict__ goals,                   // [S,N,4] x,y,heading,speed
                                                          const unsigned char* __restrict__ eval_mask,        // [S,N] or null
                                                          const double* __restrict__ edges,                   // lin[201] ang[201] accel[acc_bins+1] nd[201]
                                                          double* __restrict__ out) {
  __shared__ double s_cnt[64], s_coll[64], s_off[64];
  const int s = blockIdx.x, v = threadIdx.x, N = p.N, T1 = p.T1;
  const double* e_lin = edges;
  const double* e_ang = edges + 201;
  const double* e_acc = edges + 402;
  const double* e_nd = edges + 402 + p.acc_bins + 1;
  double has = 0.0, c_any = 0.0, o_any = 0.0;
  if (v < N && (!eval_mask || eval_mask[(size_t)s * N + v])) {
    const float* st = states + ((size_t)s * N + v) * T1 * 8;
    const unsigned char* cl = coll + ((size_t)s * N + v) * T1 * 2;
    const double* g = gt + ((size_t)s * N + v) * T1 * 5;
    const double gx = goals[((size_t)s * N + v) * 4], gy = goals[((size_t)s * N + v) * 4 + 1];
    int first = -1, last = -1, cnt = 0;
    for (int t = p.hist_steps; t < T1; ++t)
      if (st[t * 8 + 7] != 0.f) { if (first < 0) first = t; last = t; ++cnt; }
    if (cnt > 0) {
      bool latch = false, goal = false, cany = false, oany = false;
      double err_sum = 0.0, fde = 0.0;
      for (int t = 0; t < T1; ++t) {
        const double x = (double)st[t * 8], y = (double)st[t * 8 + 1];
        // reward[0]: 1 from the first step on at which the vehicle is within the tolerance of its goal (utils/sim.py:99-104)
        const double dxg = gx - x, dyg = gy - y;
        if (sqrt(dxg * dxg + dyg * dyg) < p.pos_tol) latch = true;
        const bool m = t >= p.hist_steps && st[t * 8 + 7] != 0.f;
        if (!m) continue;
        goal = goal || latch;
        cany = cany || cl[t * 2] == 1;
        oany = oany || cl[t * 2 + 1] == 1;
        const double ex = x - g[t * 5], ey = y - g[t * 5 + 1];
        const double err = sqrt(ex * ex + ey * ey);
        err_sum += err;
        if (t == last) fde = err;
        const double vx = (double)st[t * 8 + 2], vy = (double)st[t * 8 + 3];
        hist_add(out + M_LIN_SIM, e_lin, 200, clipd(sqrt(vx * vx + vy * vy), 0.0, 30.0));
        hist_add(out + M_LIN_GT, e_lin, 200, clipd(g[t * 5 + 3], 0.0, 30.0));
        hist_add(out + M_ANG_SIM, e_ang, 200, clipd((double)st[t * 8 + 4] / p.dt, -50.0, 50.0));
        hist_add(out + M_ANG_GT, e_ang, 200, clipd(g[t * 5 + 2] / p.dt, -50.0, 50.0));
        if (t != first && t != last) {
          // log acceleration: central difference for 0 < t < steps - 1, else 0; discretised like the actions (:283-289)
          double ga = (t > 0 && t < T1 - 2) ? (g[(t + 1) * 5 + 3] - g[(t - 1) * 5 + 3]) / (2 * p.dt) : 0.0;
          ga = (clipd(ga, p.min_accel, p.max_accel) - p.min_accel) / (p.max_accel - p.min_accel);
          ga = rint(ga * (p.n_accel - 1)) / (p.n_accel - 1);
          ga = ga * (p.max_accel - p.min_accel) + p.min_accel;
          hist_add(out + M_ACC_GT, e_acc, p.acc_bins, ga);
          // applied acceleration of step t: the centre of the sampled token's acceleration bin (0 after the last step)
          double sa = 0.0;
          if (t < T1 - 1) sa = (double)(tok[((size_t)s * N + v) * p.Tmax + t] / p.n_steer) / (double)(p.n_accel - 1) *
                                (p.max_accel - p.min_accel) + p.min_accel;
          hist_add(out + M_ACC_SIM, e_acc, p.acc_bins, sa);
        }
        // nearest existing vehicle, in the rollout and in the log (existence of the ROLLOUT masks both)
        double best = __builtin_inf(), best_gt = __builtin_inf();
        for (int j = 0; j < N; ++j) {
          if (j == v) continue;
          const float* sj = states + (((size_t)s * N + j) * T1 + t) * 8;
          if (sj[7] == 0.f) continue;
          const double dx = x - (double)sj[0], dy = y - (double)sj[1];
          best = fmin(best, dx * dx + dy * dy);
          const double* gj = gt + (((size_t)s * N + j) * T1 + t) * 5;
          const double hx = g[t * 5] - gj[0], hy = g[t * 5 + 1] - gj[1];
          best_gt = fmin(best_gt, hx * hx + hy * hy);
        }
        hist_add(out + M_ND_SIM, e_nd, 200, clipd(best == __builtin_inf() ? 0.0 : sqrt(best), 0.0, 40.0));
        hist_add(out + M_ND_GT, e_nd, 200, clipd(best_gt == __builtin_inf() ? 0.0 : sqrt(best_gt), 0.0, 40.0));
      }
      atomicAdd(&out[M_SUMS + 0], goal ? 1.0 : 0.0);
      atomicAdd(&out[M_COUNTS + 0], 1.0);
      atomicAdd(&out[M_SUMS + 3], err_sum / cnt);
      atomicAdd(&out[M_COUNTS + 3], 1.0);
      atomicAdd(&out[M_SUMS + 4], fde);
      atomicAdd(&out[M_COUNTS + 4], 1.0);
      has = 1.0; c_any = cany ? 1.0 : 0.0; o_any = oany ? 1.0 : 0.0;
    }
  }
  s_cnt[v] = has; s_coll[v] = c_any; s_off[v] = o_any;
  __syncthreads();
  if (v == 0) {                                        // per-scenario collision / off-road RATES over its evaluated vehicles (:246-248)
    double n = 0.0, c = 0.0, o = 0.0;
    for (int i = 0; i < 64; ++i) { n += s_cnt[i]; c += s_coll[i]; o += s_off[i]; }
    if (n > 0.0) {
      atomicAdd(&out[M_SUMS + 1], c / n); atomicAdd(&out[M_COUNTS + 1], 1.0);
      atomicAdd(&out[M_SUMS + 2], o / n); atomicAdd(&out[M_COUNTS + 2], 1.0);
    }
  }
}

extern "C" int ctrlsim_metrics_size(void) { return M_TOTAL; }
extern "C" int ctrlsim_metrics_pack(int S, int N, int T1, int Tmax, int hist_steps, double dt, const float* hist_states,
                                    const unsigned char* coll, const int* hist_tok, const double* gt, const double* goals4,
                                    const unsigned char* eval_mask, const double* params5, const double* edges, double* out,
                                    hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (N < 1 || N > 64 || T1 < 2 || !hist_states || !coll || !hist_tok || !gt || !goals4 || !params5 || !edges || !out)
    return CTRLSIM_EINVAL;
  MetricParams p;
  p.N = N; p.T1 = T1; p.Tmax = Tmax; p.hist_steps = hist_steps; p.dt = dt;
  p.pos_tol = params5[0]; p.min_accel = params5[1]; p.max_accel = params5[2];
  p.n_accel = (int)params5[3]; p.n_steer = (int)params5[4];
  p.acc_bins = 20;
  hipLaunchKernelGGL(metrics_pack_kernel, dim3(S), dim3(64), 0, st, p, hist_states, coll, hist_tok, gt, goals4, eval_mask, edges, out);
  return ctrlsim_launch_status();
}
