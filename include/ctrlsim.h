/* ctrlsim.h — C ABI of the MI355X-native CtRL-Sim closed-loop rollout path (libctrlsim_hip.so).
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (in this repo: torch tensors, data_ptr()),
 * contiguous, little-endian; every function takes a hipStream_t, enqueues work on it and returns immediately
 * (0 = OK, negative errno-style code otherwise: -22 invalid argument, -5 launch failure); nothing allocates,
 * synchronises or throws.  No torch types appear here: the library links only against the HIP runtime.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference repository
 * montrealrobotics/ctrl-sim @ 2024_10_08).  The pybind11 module `nocturne_cpp` (the .cc files under nocturne/pybind11/src) and the
 * per-vehicle Python dict traffic of evaluators/policy_evaluator.py:99-159,514-542 are what a host binding calls today;
 * INTEGRATION.md shows the ctypes stub a maintainer adds to call these instead.
 *
 * Layouts (S scenarios, N <= 64 vehicles each, A context slots, T context steps, Tq = token_index+1 <= T,
 * P polylines x NP points per context, P_all polylines per scenario, Tmax rollout steps, Tmax1 = Tmax+1):
 *   hist_states [S,N,Tmax1,8] f32   x, y, vx, vy, heading, length, width, existence   (policies/policy.py:68-79)
 *   hist_tok    [S,N,Tmax]    i32   applied action token per step (default ZERO_ACTION_TOKEN = 524)
 *   hist_rtg    [S,N,Tmax,3]  i32   sampled RTG bins per step (default (0,35,35))
 *   coll        [S,N,Tmax1,2] u8    vehicle-vehicle / vehicle-road-edge collision flags
 *   phys        [S,N,20]      f32   Box2D body + FreeCar control state (see csrc/sim.hip)
 */
#ifndef CTRLSIM_H
#define CTRLSIM_H

#include <stdint.h>

/* context size classes a model batch / class-table launch may hold (csrc/common.h: MAXC) */
#define CTRLSIM_MAX_CLASSES 16

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

typedef struct ctrlsim_dims {
  int A, T, P, NP, D, H, F, V, R, C, NE, ND, MAXT;
  int variant; /* 0 CtRL-Sim (state, rtg, action tokens; cfgs/model/ctrl_sim.yaml), 1 IL (state, action; il.yaml),
                  2 Trajeglish (action tokens only; trajeglish.yaml), 3 Decision Transformer (continuous RTGs: ctx.rtg_bin
                  holds float bits; token order rtg, state, action; dt.yaml) — modules/encoder.py:27-34,116-152,
                  decoder.py:29-64; 4 (round 6) = the CtRL-Sim tokens and heads of 0 under cfg.model.attend_own_return_action
                  (cfgs/model/base.yaml:15, utils/train_utils.py:114-129: other agents' return / action tokens of EARLIER
                  timesteps are hidden; attention mask mode 5; plain 24-slot contexts, no K/V-cached entry point) */
  int flags;   /* model switches of cfgs/model/base.yaml beyond the shipped values (round 6; 0 = shipped; CtRL-Sim token layout only):
                  1 = no_actions (modules/encoder.py:129-130: action embeddings x 0 in front of embed_ln),
                  2 = use_map False (encoder.py:18,155-170: no MapEncoder — the polyline rows of the scene memory are key-padded, the weights
                      named encoder.map_encoder.* may be zeros),
                  4 = encode_initial_state False (encoder.py:84,111,135,159-166: the initial-state rows are key-padded).  2 and 4 together are
                      refused (the reference has no scene memory to build then). */
} ctrlsim_dims;

/* Agent-local context tensors of B contexts (outputs of ctrlsim_build_context, inputs of the forward). */
typedef struct ctrlsim_ctx {
  float* st12;       /* [B,Tq,A,12] x,y,vx,vy,yaw,len,wid + 5 type one-hot (-1 on padded slots) */
  float* exist;      /* [B,Tq,A]                                                                 */
  float* goal5;      /* [B,A,5]                                                                  */
  int* act_tok;      /* [B,Tq,A]                                                                 */
  int* rtg_bin;      /* [B,Tq,A,3]                                                               */
  int* tstep;        /* [B,Tq]                                                                   */
  int* slot_gid;     /* [B,A] global vehicle index per slot, -1 = padding                        */
  float* road_pts;   /* [B,P,NP,3] x,y,exist in the focal frame                                  */
  float* road_types; /* [B,P,8] one-hot, -1 rows = padding                                       */
} ctrlsim_ctx;

typedef struct ctrlsim_model ctrlsim_model;

/* ---- simulator step -------------------------------------------------------------------------------------------
 * Replaces nocturne_cpp Simulation.step(dt) + the per-vehicle setters (nocturne/pybind11/src/simulation.cc:20-38,
 * object.cc:52-54,89-92, vehicle.cc:20) i.e. Vehicle::set_acceleration/brake/set_steering/set_position,
 * PhysicsSimulation::Step -> FreeCar::Step + b2World::Step, Vehicle::Step, Scenario::UpdateCollision
 * (nocturne/cpp/src/vehicle.cc:25-135, physics/FreeCar.cpp:66-186, scenario.cc:266-328), AutoregressivePolicy.act
 * (policies/autoregressive_policy.py:256-274) and the state read-back of update_vehicle_data_dict
 * (evaluators/policy_evaluator.py:99-121).
 * contact_state: [S, ctrlsim_sim_contact_floats(N)] floats owned by the caller, initialised by ctrlsim_sim_init and carried from
 * step to step: Box2D's persistent per-pair contact manifolds with their accumulated impulses (warm starting),
 * b2World::m_inv_dt0 and the broad phase (fat AABBs, move buffer, b2DynamicTree nodes).  With it the step includes Box2D's box-box contact handling between vehicles (b2CollidePolygons,
 * islands, b2ContactSolver: third_party/box2d/src/collision/b2_collide_polygon.cpp, src/dynamics/b2_contact_solver.cpp,
 * b2_island.cpp, b2_world.cpp); NULL = contact-free integration (vehicles pass through each other; flags still exact). */
int64_t ctrlsim_sim_contact_floats(int N);
int ctrlsim_sim_init(int S, int N, int E, const float* init_pose /*[S,N,4] x,y,heading,speed*/,
                     const float* size /*[S,N,2] length,width*/, const float* edges /*[S,E,4]*/,
                     const uint8_t* exists /*[S,N]*/, float* phys, float* hist_states, uint8_t* coll, int Tmax1,
                     float* contact_state, hipStream_t stream);
/* Object.setPosition(x, y) / set_position (nocturne/pybind11/src/object.cc:52-54,87-90 -> Vehicle::set_position, vehicle.cc:75-87 ->
 * BaseCar::SetPosition = b2Body::SetTransform at the current angle, physics/BaseCar.cpp:28-32): xy [S,N,2], NaN = leave the vehicle
 * where it is.  The request is parked in `phys` and applied at the top of the next ctrlsim_sim_step, before that step's controls
 * (the reference applies it immediately; nothing observes the body in between).  The rollout's own use of it — vehicles that
 * stopped existing are parked at (-1e6, -1e6) every step, autoregressive_policy.py:260-263 — is what `exists` = 0 does. */
int ctrlsim_sim_set_position(int S, int N, const float* xy, float* phys, hipStream_t stream);
/* act_tok [S,N] (token id, or -1 = zero action) or act_f64 [S,N,2] (accel, steer); disc6 = {min_accel, max_accel,
 * min_steer, max_steer, n_accel, n_steer}; applied (nullable) [S,N,2] f64 receives the continuous actions.
 * mode 0 = FreeCar/Box2D (what eval_sim.py executes), 1 = Object::KinematicBicycleStep (object.cc:126-137). */
int ctrlsim_sim_step(int S, int N, int E, const int* act_tok, const double* act_f64, const double* disc6,
                     const float* size, const float* edges, const uint8_t* exists, float* phys, float* hist_states,
                     uint8_t* coll, double* applied, int t, int Tmax1, float dt, int mode, float* contact_state,
                     hipStream_t stream);
/* The same step (mode 0) with EXPERT-CONTROLLED objects (Object.expert_control = True, nocturne/pybind11/src/object.cc:58-59;
 * Scenario::Step, nocturne/cpp/src/scenario.cc:272-284; utils/sim.py:20-65 get_ground_truth_states): expert [S,N,4] = logged x, y,
 * heading, speed of step t + 1, x = NaN for the vehicles under policy control.  The world step moves every body; an expert vehicle is then
 * put on its logged state through Vehicle::set_position / set_heading / set_speed (vehicle.cc:75-105: two b2Body::SetTransform with proxy
 * synchronisation and the new-contact search of the next step, SetLinearVelocity) before the collision flags are taken; its history
 * row holds the logged values.  expert == NULL: ctrlsim_sim_step. */
int ctrlsim_sim_step_expert(int S, int N, int E, const int* act_tok, const double* act_f64, const double* disc6,
                            const float* size, const float* edges, const uint8_t* exists, float* phys, float* hist_states,
                            uint8_t* coll, double* applied, int t, int Tmax1, float dt, float* contact_state,
                            const float* expert, hipStream_t stream);

/* ---- focal grouping + context tensors ------------------------------------------------------------------------
 * Replaces AutoregressivePolicy.get_data (policies/autoregressive_policy.py:51-165) with
 * RLWaymoDataset.select_relevant_agents / normalize_scene (datasets/rl_waymo/dataset.py:278-319,390-428). */
int ctrlsim_group_build(int S, int N, int A, int T, int t, int Tmax1, double dist_thresh, const float* hist_states,
                        const int* eval_order /*[S,N] -1 padded*/, int has_roads, uint64_t* persist /*[S,N]*/,
                        int* n_groups /*[S]*/, int* grp_focal /*[S,N]*/, uint64_t* grp_ids, uint64_t* grp_members,
                        int* own_g /*[S,N]*/, int* mem_g /*[S,N]*/, uint8_t* tilted /*[S,N]*/, hipStream_t stream);
/* flag[0] |= 1 if the focal groups of scenarios [0,S) (count, focal vehicle or membership mask of any group) differ from the
 * snapshot ref_*: the device-side form of "did get_data build the same contexts as last step" that gates the K/V-cached
 * phase; the caller zeroes flag and copies it to the host together with n_groups (one asynchronous copy per step). */
int ctrlsim_groups_changed(int S, int N, const int* n_groups, const int* grp_focal, const uint64_t* grp_ids, const int* ref_n,
                           const int* ref_focal, const uint64_t* ref_ids, int* flag, hipStream_t stream);
/* Compact contexts.  Token rows of agent slots that exist nowhere in the window are identical (modules/encoder.py:127-133 multiplies
 * every embedding by the existence flag before embed_ln; the decoder puts no key padding on its targets) and remain identical
 * through all layers, so a context with n vehicles may be evaluated with any slot count Actx >= n + 1: Actx - 1 regular slots and
 * ONE representative slot standing for the dims.A - (Actx - 1) padded slots of the reference layout (its keys carry that
 * multiplicity in the attention: ctrlsim_attention_compact).  Exact in real arithmetic.  `sizes` = the ascending slot counts a
 * caller uses (nb <= CTRLSIM_MAX_CLASSES, last = dims.A; a context takes the first size >= n + 1, or dims.A).
 * ctrlsim_group_size_hist: hist[s, k] = focal groups of scenario s in size class k (with n_groups: what the host needs to cut a
 * step into model batches).  ctrlsim_ctx_index_classes: ctrlsim_ctx_index with the contexts SORTED by size class (class k =
 * contexts [sum_{j<k} count_j, ...)); ctx_row0[c] = first logits row of context c when class k emits (sizes[k] - 1, or dims.A
 * for the last class) rows per context; ctx_of_group: [S, N] scratch. */
int ctrlsim_group_size_hist(int S, int N, const int* n_groups, const uint64_t* grp_ids, int nb, const int* sizes /*host*/,
                            int* hist /*[S,nb]*/, hipStream_t stream);
int ctrlsim_ctx_index_classes(int s0, int s1, int N, int A, const int* n_groups, const uint64_t* grp_ids, const int* own_g,
                              const int* mem_g, int nb, const int* sizes /*host*/, int* ctx_scn, int* ctx_grp, int* ctx_row0,
                              int* ctx_of_group, int* own_ctx, int* own_slot, int* mem_ctx, int* mem_slot, hipStream_t stream);
int ctrlsim_ctx_index(int s0, int s1, int N, const int* n_groups, const int* grp_focal, const uint64_t* grp_ids,
                      const int* own_g, const int* mem_g, int* ctx_scn, int* ctx_grp, int* own_ctx, int* own_slot,
                      int* mem_ctx, int* mem_slot, int* ctx_base, hipStream_t stream);
int ctrlsim_build_context(int B, int N, int A, int T, int t, int Tq, int tt_first /*emit window rows [tt_first,Tq)*/,
                          int Tmax1, int Tmax, int P_all, int P, int NP,
                          const int* ctx_scn, const int* ctx_grp, const int* grp_focal, const uint64_t* grp_ids,
                          const float* hist_states, const int* hist_tok, const int* hist_rtg,
                          const double* goals /*[S,N,5]*/, const float* types /*[S,N,5]*/,
                          const float* roads /*[S,P_all,NP,3]*/, const float* road_types /*[S,P_all,8]*/,
                          const int* zero4 /*{zero action token, rtg bins x3}*/, const ctrlsim_ctx* out,
                          hipStream_t stream);
/* The same for a model batch of up to CTRLSIM_MAX_CLASSES context size classes in ONE launch (class k: B[k] contexts of A[k] slots written to
 * out[k]; the batch's context list ctx_scn / ctx_grp holds the classes back to back). */
int ctrlsim_build_context_c(int n, const int* B, const int* A, const ctrlsim_ctx* out, int N, int T, int t, int Tq, int tt_first,
                            int Tmax1, int Tmax, int P_all, int P, int NP, const int* ctx_scn, const int* ctx_grp,
                            const int* grp_focal, const uint64_t* grp_ids, const float* hist_states, const int* hist_tok,
                            const int* hist_rtg, const double* goals, const float* types, const float* roads,
                            const float* road_types, const int* zero4, hipStream_t stream);

/* ---- model ----------------------------------------------------------------------------------------------------
 * Replaces CtRLSim.load_from_checkpoint + CtRLSim.forward (models/ctrl_sim.py:19-45; modules/encoder.py,
 * map_encoder.py, decoder.py) as used by AutoregressivePolicy.predict (autoregressive_policy.py:189-210).
 * `names`/`offsets` describe the packed fp32 weight buffer (state_dict names + the "fold.*" tensors of
 * ctrlsim_amd/pack.py). */
int ctrlsim_model_create(const ctrlsim_dims* dims, const float* dev_weights, int n, const char* const* names,
                         const int64_t* offsets, ctrlsim_model** out);
void ctrlsim_model_destroy(ctrlsim_model* m);
int64_t ctrlsim_forward_workspace_bytes(const ctrlsim_dims* dims, int B, int Tq);
/* The *_a forms take a uniform batch of COMPACT contexts of Actx slots (context tensors [B,Tq,Actx,.]; Actx == dims.A: the plain
 * layout, what the forms without _a call): Actx - 1 regular slots + the representative (see ctrlsim_group_size_hist); logits
 * come back for the regular slots only, [B*(Actx-1), .] rows (Actx == dims.A: [B*A, .]). */
int64_t ctrlsim_forward_workspace_bytes_a(const ctrlsim_dims* dims, int B, int Tq, int Actx);
int ctrlsim_dt_forward_pass1_a(const ctrlsim_model* m, int B, int Tq, int Actx, const ctrlsim_ctx* ctx, void* workspace,
                               float* rtg_logits, float* dbg_seg_emb, hipStream_t stream);
int ctrlsim_dt_forward_pass2_a(const ctrlsim_model* m, int B, int Tq, int Actx, int t, int N, int Tmax, const ctrlsim_ctx* ctx,
                               const int* ctx_scn, const int* hist_rtg, void* workspace, float* act_logits, int cached,
                               hipStream_t stream);
int ctrlsim_dt_forward_pass1_cached_a(const ctrlsim_model* m, int B, int t, int Actx, const ctrlsim_ctx* ctx, void* workspace,
                                      float* rtg_logits, hipStream_t stream);
/* The *_c forms take a model BATCH of n <= CTRLSIM_MAX_CLASSES classes of compact contexts (class k: B[k] contexts of A[k] slots, context tensors
 * ctx[k]; host arrays): every row-wise kernel runs once over the rows of all classes, the attention kernel and the K/V-image
 * epilogue of the QKV projection work from a class table.  Logits rows come class after class, context after context,
 * regular slot after regular slot (ctrlsim_ctx_index_classes' ctx_row0).  ctx_scn lists the contexts in that same order. */
int64_t ctrlsim_forward_workspace_bytes_c(const ctrlsim_dims* dims, int n, const int* B, const int* A, int Tq);
int ctrlsim_dt_forward_pass1_c(const ctrlsim_model* m, int n, const int* B, const int* A, const ctrlsim_ctx* ctx, int Tq,
                               void* workspace, float* rtg_logits, float* dbg_seg_emb, hipStream_t stream);
int ctrlsim_dt_forward_pass2_c(const ctrlsim_model* m, int n, const int* B, const int* A, const ctrlsim_ctx* ctx, int Tq, int t,
                               int N, int Tmax, const int* ctx_scn, const int* hist_rtg, void* workspace, float* act_logits,
                               int cached, hipStream_t stream);
/* ctrlsim_dt_forward_pass1_c with its few-row tail (last decoder layer on the queried rows, the head) on `tail_stream`, ordered
 * behind the full-row part on `stream` by an event inside the call: the caller's next full-row work on `stream` need not wait
 * for it; rtg_logits are complete in tail_stream order. */
int ctrlsim_dt_forward_pass1_c2(const ctrlsim_model* m, int n, const int* B, const int* A, const ctrlsim_ctx* ctx, int Tq,
                                void* workspace, float* rtg_logits, hipStream_t stream, hipStream_t tail_stream);
int ctrlsim_dt_forward_pass1_cached_c(const ctrlsim_model* m, int n, const int* B, const int* A, const ctrlsim_ctx* ctx, int t,
                                      void* workspace, float* rtg_logits, hipStream_t stream);
/* pass 1: rtg_logits [B,A,R*C] of the current-timestep state tokens; caches per-layer K/V in `workspace`. */
int ctrlsim_dt_forward_pass1(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* ctx, void* workspace,
                             float* rtg_logits, float* dbg_seg_emb /*nullable [B,P,D]*/, hipStream_t stream);
/* The baselines of cfgs/model/{il,trajeglish,dt}.yaml (dims.variant 1 / 2 / 3) have no RTG head and one forward per step
 * (policies with predict_rtgs = False, autoregressive_policy.py:189,208-240): action logits [B,A,V] of the current step from
 * the state tokens (IL, DT) / action tokens (Trajeglish), decoder.py:55-64.  ctrlsim_dt_forward_pass1 / _pass2 / _cached refuse
 * these models and this call refuses the CtRL-Sim model. */
int ctrlsim_dt_forward_actions(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* ctx, void* workspace,
                               float* act_logits, hipStream_t stream);
/* Component-level entry (tests, micro-benchmarks): MapEncoder's point MLP + seed-attention pooling (modules/map_encoder.py:28-46,
 * before out_proj) of B*P polylines road_pts [B,P,NP,3] -> attn_pre [B*P,256]; pad [B,P] <- 1 for polylines without any point. */
int ctrlsim_map_pool(const ctrlsim_model* m, int B, const float* road_pts, float* attn_pre, uint8_t* pad, hipStream_t stream);
/* The reference's full return contract of CtRLSim.forward (models/ctrl_sim.py:41-45; decoder.py:52-77): teacher-forced, every
 * head on every token of the Tq window steps.  action_preds [B,Tq,A,V], rtg_preds [B,Tq,A,R*C], state_preds [B,Tq,A,2T] in
 * token-row order (the reference returns the [B,A,T,.] permutation of these).  rtg_preds / state_preds are nullable and must be
 * NULL for models without the head (predict_rtg / predict_future_states false: il, trajeglish, dt yaml).  Workspace as pass 1.
 * Training-time / debugging contract — the rollout reads one timestep and uses the pass1 / pass2 entry points. */
int ctrlsim_forward_all(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* ctx, void* workspace, float* action_preds,
                        float* rtg_preds, float* state_preds, hipStream_t stream);
/* pass 2 (same workspace, after ctrlsim_sample_rtg wrote hist_rtg[...,t,:]): act_logits [B,A,V].
 * cached = 1 pairs with ctrlsim_dt_forward_pass1_cached (workspace sized with Tq = T, ctx = last min(Tq,2) window rows). */
int ctrlsim_dt_forward_pass2(const ctrlsim_model* m, int B, int Tq, int t, int N, int Tmax, const ctrlsim_ctx* ctx,
                             const int* ctx_scn, const int* hist_rtg, void* workspace, float* act_logits, int cached,
                             hipStream_t stream);
/* pass 1 while the window still starts at step 0 (t < T) for a FIXED set of contexts: the scene side is evaluated at
 * t == 0 only and decoder K/V are cached per layer in `workspace` (ctrlsim_forward_workspace_bytes(dims, B, T), the same
 * buffer at every step); step t re-evaluates the 4A rows whose inputs changed.  ctx holds the window rows
 * [max(t-1,0), t] (ctrlsim_build_context with tt_first = max(t-1,0)). */
int ctrlsim_dt_forward_pass1_cached(const ctrlsim_model* m, int B, int t, const ctrlsim_ctx* ctx, void* workspace,
                                    float* rtg_logits, hipStream_t stream);

/* ---- sampling -------------------------------------------------------------------------------------------------
 * Replaces Policy.process_predicted_rtg (policies/policy.py:108-142) and the action-sampling block of
 * AutoregressivePolicy.predict (autoregressive_policy.py:211-240): tilt, softmax, torch.multinomial as an
 * exponential race with explicit (noise != NULL) or in-kernel counter-based Exp(1) noise.
 * tilt3 = (goal, veh_veh, veh_edge) tilts of the policy (host pointer; tilt_dict of policies/policy.py:20-25) applied to
 * every scenario, unless tilt_scn (device, [S,3]) gives one triple per scenario — a reward-tilt sweep in one batch. */
int ctrlsim_sample_rtg(const float* rtg_logits, int A, int R, const int* own_ctx, const int* own_slot,
                       const uint8_t* tilted, const double* tilt3, const double* tilt_scn /*[S,3] or NULL*/,
                       const float* noise /*[S*N,3,R] or NULL*/, uint64_t seed, const int64_t* scenario_id /*[S]*/, int t,
                       int* hist_rtg, int S, int N, int Tmax, hipStream_t stream);
/* The same with logits rows addressed through ctx_row0 (context c, slot s -> row ctx_row0[c] + s): batches of compact contexts
 * of different slot counts (ctrlsim_ctx_index_classes). */
int ctrlsim_sample_rtg_rows(const float* rtg_logits, const int* ctx_row0, int R, const int* own_ctx, const int* own_slot,
                            const uint8_t* tilted, const double* tilt3, const double* tilt_scn, const float* noise, uint64_t seed,
                            const int64_t* scenario_id, int t, int* hist_rtg, int S, int N, int Tmax, hipStream_t stream);
int ctrlsim_sample_action_rows(const float* act_logits, const int* ctx_row0, int V, const int* mem_ctx, const int* mem_slot,
                               float temperature, double top_p, const float* noise, uint64_t seed, const int64_t* scenario_id,
                               int t, int* hist_tok, int* act_now, int S, int N, int Tmax, int zero_token, hipStream_t stream);
int ctrlsim_sample_action(const float* act_logits, int A, int V, const int* mem_ctx, const int* mem_slot,
                          float temperature, double top_p /*<=0: off*/, const float* noise /*[S*N,V] or NULL*/,
                          uint64_t seed, const int64_t* scenario_id, int t, int* hist_tok, int* act_now /*[S,N]*/,
                          int S, int N, int Tmax, int zero_token, hipStream_t stream);

/* ---- real-time reward ledger (Decision-Transformer baseline) --------------------------------------------------------
 * Replaces the host bookkeeping of a policy with real_time_rewards (cfgs/policy/dt.yaml): PolicyEvaluator.update_vehicle_data_dict
 * (evaluators/policy_evaluator.py:122-147: RTG_0, RTG_t = RTG_{t-1} - dense_reward_{t-1}) + Evaluator.compute_dense_reward
 * (evaluators/evaluator.py:106-140; datasets/rl_waymo/dataset.py:187-275; utils/data.py:152-290) + the clip-normalisation of
 * AutoregressivePolicy.get_data (policies/autoregressive_policy.py:73-78).  Call once per step t, after the simulator wrote the
 * states of step t and before the contexts of step t are built: hist_rtg[S,N,Tmax,3][.., t, :] <- float bits of the normalised
 * RTG (what a variant-3 model reads); `ledger` [S,N,10] float64 carries RTG / dense reward / step-0 reward flags between calls
 * (t = 0 initialises it); init_rtg [S,N,3] float64 or NULL = (10, 90, 90) (max_return); rtg_raw [S,N,Tmax,3] float64 nullable.
 * `edges` is the simulator's road-edge segment table [S,E,4]. */
typedef struct ctrlsim_dt_reward_cfg {
  double pos_tol;        // cfg.nocturne.rew_cfg.position_target_tolerance
  double shaped_unit;    // shaped_goal_distance_scaling / reward_scaling (utils/sim.py:112-118)
  double goal_mult, shaped_min, shaped_max;     // pos_target_achieved_rew_multiplier, pos_goal_shaped_min / _max
  double veh_mult, max_veh_dist;                // veh_veh_collision_rew_multiplier, max_veh_veh_distance
  double edge_mult, edge_scale;                 // veh_edge_collision_rew_multiplier, dist_to_road_edge_scaling_factor
  double rtg_lo[3], rtg_hi[3];                  // min / max_rtg_pos, _veh, _road
  int remove_shaped_goal, remove_shaped_veh, remove_shaped_edge, pad_;
} ctrlsim_dt_reward_cfg;
int ctrlsim_dt_ledger_step(int S, int N, int E, int t, int T1, int Tmax, const float* hist_states, const uint8_t* coll,
                           const double* goals /*[S,N,5]: x, y first*/, const float* edges, const double* init_rtg,
                           const ctrlsim_dt_reward_cfg* cfg, double* ledger, double* rtg_raw, int* hist_rtg, hipStream_t stream);

/* ---- metrics ----------------------------------------------------------------------------------------------------
 * Replaces PolicyEvaluator.update_running_statistics (evaluators/policy_evaluator.py:162-248) with compute_reward's goal latch
 * (utils/sim.py:99-104) and compute_nearest_dist_all (evaluators/evaluator.py:87-103) for S finished rollouts: out[0 ..
 * ctrlsim_metrics_size()) += the packed accumulators — sums and counts of (goal, per-scenario collision rate, per-scenario
 * off-road rate, ADE, FDE) and the eight histograms compute_metrics builds (policy_evaluator.py:251-305), in the order of
 * ctrlsim_amd/metrics.py:MetricAccumulators.pack.  This vector is what the ranks all-reduce (the only collective of a
 * multi-GPU run).  gt [S,N,T1,5] = logged x, y, heading, speed, exist; goals4 [S,N,4] = goal x, y, heading, speed (after
 * initialize_goal_dict); eval_mask [S,N] (NULL = all vehicles evaluated); params5 (host) = position tolerance, min_accel,
 * max_accel, n_accel_bins, n_steer_bins; edges (device) = the histogram edges lin[201] ang[201] accel[21] nearest[201].  The applied
 * acceleration of a step is the centre of the sampled token's acceleration bin. */
int ctrlsim_metrics_size(void);
int ctrlsim_metrics_pack(int S, int N, int T1, int Tmax, int hist_steps, double dt, const float* hist_states, const uint8_t* coll,
                         const int* hist_tok, const double* gt, const double* goals4, const uint8_t* eval_mask,
                         const double* params5, const double* edges, double* out, hipStream_t stream);

/* ---- building blocks (exported for parity tests and profiling) ------------------------------------------------ */
int ctrlsim_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr,
                    float* C, int ldc, int M, int N, int K, int relu, hipStream_t stream);
/* Same product evaluated on the bf16 MFMA with fp32-class accuracy: W3 = the weight split into three bf16 planes, slab-major
 * [K/16][3][2][n_total][8] (ctrlsim_amd/pack.py:split3_planes); rows [n0, n0+N) of it are used; A stays fp32.
 * ln_gamma/ln_beta non-NULL (N must be 256): C = [relu] LayerNorm(A W^T + bias [+ R]) * gamma + beta, eps 1e-5 — the
 * post-LN residual blocks of nn.TransformerEncoder/DecoderLayer and the Linear-LayerNorm-ReLU halves of MLPLayer
 * (modules/layers.py) as one kernel; C may alias R. */
int ctrlsim_gemm_nt_bf16x6(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, const float* R,
                           int ldr, float* C, int ldc, int M, int N, int K, int relu, const float* ln_gamma,
                           const float* ln_beta, hipStream_t stream);
/* Plain Linear 256 -> 256 (W3 rows [n0, n0 + 256), K = 256, two-fp16-plane scheme) whose result row i is written to row c_rows[i] of C
 * (device int32 list of M entries; an entry < 0 is not stored): the map encoder's last Linear (modules/map_encoder.py:53 feeding
 * modules/encoder.py:155-158's concatenation) writes the polyline rows of the scene-encoder source [b, 0..P-1] itself, no copy kernel.
 * Returns 1 — nothing launched — when the weight-stationary kernel does not apply (bf16x6 split selected, option 6 off): the caller
 * runs ctrlsim_gemm_nt_bf16x6 + a row copy instead. */
int ctrlsim_gemm256_rows(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, float* C, int ldc,
                         const int* c_rows, int M, hipStream_t stream);
/* The same Linear whose last 512 output columns [kv_col0, kv_col0 + 512) are the keys / values of 8 heads x 32 of contexts of
 * kv_L rows each: those columns leave the epilogue directly as the split K / V tile images of ctrlsim_kv_split (kv_nkt 64-key
 * tiles per context and head; the tail of the last tile must be zeroed by the caller), columns [0, kv_col0) as fp32 rows of C.
 * The in_proj of nn.MultiheadAttention (modules/decoder.py:16-20, encoder.py:42-46) fused with the attention kernel's operand
 * preparation. */
int ctrlsim_gemm_nt_kv(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, float* C, int ldc,
                       int M, int N, int K, void* kv_img, int kv_L, int kv_nkt, int kv_col0, hipStream_t stream);
/* The same operation through the row-stationary kernel (two-fp16-plane scheme only; CTRLSIM_EINVAL otherwise): Wblk = the weight as
 * 32-column operand blocks (ctrlsim_amd/pack.py:row_blocks of rows [n0, n0 + N) of the weight; N = kv_col0 + 512, a multiple of 32, at
 * most 768), bias = its N entries; K = 256.  Every activation row is read once; outputs as for ctrlsim_gemm_nt_kv.  kv_img == NULL: a
 * plain Linear (N a multiple of 32 in [64, 768]; kv_L / kv_nkt / kv_col0 ignored), all N columns as fp32 rows of C. */
int ctrlsim_gemm_kv_blocks(const float* A, int lda, const void* Wblk, const float* bias, float* C, int ldc, int M, int N,
                           void* kv_img, int kv_L, int kv_nkt, int kv_col0, hipStream_t stream);
/* Post-LN feed-forward block of nn.TransformerEncoderLayer / DecoderLayer as one kernel:
 * Y = LayerNorm(X + W2 relu(W1 X + b1) + b2) * gamma + beta, rows of 256, F hidden units (multiple of 32); W1p / W2p are the
 * operand images of ctrlsim_amd/pack.py:ffn_planes; Y may alias X.  The hidden activation never touches memory. */
int ctrlsim_ffn_fused(const float* X, int ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                      const float* gamma, const float* beta, float* Y, int ldy, int M, int F, hipStream_t stream);
/* The post-LN block in front of the feed-forward block fused into it (round 5, two-fp16-plane scheme only; CTRLSIM_EINVAL otherwise):
 * X1 = LayerNorm0(R + Wo O + bo), Y = LayerNorm(X1 + W2 relu(W1 X1 + b1) + b2) — the attention out-projection with its residual and
 * LayerNorm (nn.TransformerDecoderLayer: multihead_attn.out_proj + norm2; nn.TransformerEncoderLayer: self_attn.out_proj + norm1;
 * modules/decoder.py:16-20, encoder.py:42-46) and the feed-forward block behind it as one kernel: X1 is neither written nor re-read.
 * O = attention output rows, R = residual rows (Y may alias R or O), Wop / W1q / W2p = ctrlsim_amd/pack.py:ffn_planes_pre(Wo, W1, W2). */
int ctrlsim_ffn_fused_pre(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0,
                          const float* be0, const void* W1q, const float* b1, const void* W2p, const float* b2, const float* gamma,
                          const float* beta, float* Y, int ldy, int M, int F, hipStream_t stream);
/* X1 = LayerNorm0(R + Wo O + bo) and Q = Wq X1 + bq as one kernel (round 5, two-fp16-plane scheme only): a decoder layer's
 * self_attn.out_proj + residual + norm1 and the query third of multihead_attn.in_proj behind it (nn.TransformerDecoderLayer._sa_block /
 * _mha_block; modules/decoder.py:16-20) — the row goes in once, x1 and q come out.  X1 may alias R.  Wop / Wqp =
 * ctrlsim_amd/pack.py:outproj_q_planes(Wo, Wq). */
int ctrlsim_outproj_ln_q(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0, const float* be0,
                         const void* Wqp, const float* bq, float* X1, int ldx1, float* Q, int ldq, int M, hipStream_t stream);
int ctrlsim_layernorm256(const float* X, int ldx, const float* Radd, int ldr, const float* gamma, const float* beta,
                         float* Y, int ldy, int rows, int relu, hipStream_t stream);
/* mode 0: key padding (key_pad [B,Lk], 1 = ignore); mode 1: CtRL-Sim structured causal mask (utils/train_utils.py:81-129); modes 2 / 3 / 4
 * (split-operand kernels only): the IL / Trajeglish / Decision-Transformer masks of the same function; mode 5 (round 6): mode 1 with
 * cfg.model.attend_own_return_action (:114-129) — of the earlier timesteps a query sees the state tokens and its own agent's tokens only */
int ctrlsim_attention(int mode, const float* Q, int ldq, int64_t q_batch_stride, const float* K, const float* V,
                      int ldkv, int64_t kv_batch_stride, float* O, int ldo, int64_t o_batch_stride, const int* q_pos,
                      const uint8_t* key_pad, int B, int Lq, int Lk, int A, hipStream_t stream);
/* Split-bf16 K/V images for the bf16x6 attention (csrc/attention_bf16x6.hip): per (context, head, 64-key tile) 24 KB =
 * K planes [3][d>>3][key][d&7] then V^T planes [3][key>>2][d][key&3], bf16; img holds B * 8 * nkt tiles.
 * pos == NULL: rows [0, rows) of every context are split, keys beyond are zero-filled (rows <= 64 * nkt);
 * pos != NULL: row r of context b is written at key position pos[r] (KV-cache update; other keys untouched). */
int ctrlsim_kv_split(const float* K, const float* V, int ldkv, int64_t kv_batch_stride, const int* pos, int B, int rows,
                     int nkt, void* img, hipStream_t stream);
/* ctrlsim_attention with K/V taken from such images (staged by LDS-DMA, no in-kernel split); same modes and masks. */
int ctrlsim_attention_presplit(int mode, const float* Q, int ldq, int64_t q_batch_stride, const void* img, int nkt, float* O,
                               int ldo, int64_t o_batch_stride, const int* q_pos, const uint8_t* key_pad, int B, int Lq,
                               int Lk, int A, hipStream_t stream);

/* Causal (CtRL-Sim mask) attention over the K/V images of COMPACT contexts: Lk regular keys (slots < A) in the tiles
 * [0, ceil(Lk/64)), rep_keys representative keys (3 per window step, multiplicity rep_mult) in the tiles from ceil(rep_pos0/64)
 * on; rep_pos0 >= Lk = the regular length of the full window (K/V-cache layout).  Query positions (row index or q_pos) >=
 * rep_pos0 are the representative's own tokens.  A representative key (t, k) is visible to a query of step tq iff t < tq or
 * (t == tq and k == 0), rep_mult-fold (log2(rep_mult) is added to its score); the representative's own queries also see their
 * tokens 1..kq of step tq, once.  rep_keys = 0: ctrlsim_attention_presplit mode 1. */
int ctrlsim_attention_compact(const float* Q, int ldq, int64_t q_batch_stride, const void* img, int nkt, float* O, int ldo,
                              int64_t o_batch_stride, const int* q_pos, int B, int Lq, int Lk, int A, int rep_keys, int rep_mult,
                              int rep_pos0, hipStream_t stream);

/* Visibility-mask table of one class of causal launches over the TOKEN ROWS (query row i = position i; round 4).  The structured mask of
 * utils/train_utils.py:81-129 (get_causal_mask) — and the representative's rules above — depends on (query position, key position) of the
 * class only, so the forward evaluates it once per pass instead of per (context, head, layer, query).  Layout (csrc/attention_bf16x6.hip):
 * per query group of 32 rows and per 32-key sub-tile (2 * nkt of them: the regular tiles, then the representative's) 32 x uint64 — sixteen
 * lane masks of the visible (query, key) pairs in the kernel's accumulator order, then sixteen of the representative's count-once keys.
 * Round 6: behind the mask entries the table carries the kernel's control flow — per query group the tile schedule of its 256-query block
 * (regular / representative tiles that hold a key some query of the block sees) and per (group, 64-key tile) one 32-bit word with a 2-bit
 * code per sub-tile (0 skip, 1 every key visible, 2 apply the masks, 3 masks + count-once keys); ctrlsim_attention_mask_table_bytes covers both.
 * ctrlsim_attention_tbl = ctrlsim_attention_compact with q_pos == NULL and rep_pos0 == Lk, masks taken from the table (same results). */
int64_t ctrlsim_attention_mask_table_bytes(int Lq, int nkt);
int ctrlsim_attention_mask_table(int Lq, int Lk, int A, int rep_keys, int rep_pos0, int nkt, void* tbl, hipStream_t stream);
int ctrlsim_attention_tbl(const float* Q, int ldq, int64_t q_batch_stride, const void* img, int nkt, float* O, int ldo,
                          int64_t o_batch_stride, int B, int Lq, int Lk, int A, int rep_keys, int rep_mult, const void* mask_tbl,
                          hipStream_t stream);

/* ---- measurement hooks (bench.py): HIP-event timing of every launch of ctrlsim_prof_classes() kernel classes on its own
 * launch stream — 0 GEMM (all Linear layers incl. the fused feed-forward block), 1 attention, 2 build_context, 3 assemble_tokens,
 * 4 sim_step, 5 map_pool.  enable(1) clears and starts recording; after the caller synchronised, collect() returns per class
 * the summed milliseconds, the launch count and the algorithmic FLOPs (2*M*N*K; 128 per visible (query,key) pair and head);
 * bytes() the compulsory HBM bytes (operands read once + results written once) of the same launches.  Arrays hold
 * ctrlsim_prof_classes() entries.  Process-global and single-threaded like the options below: one host thread per GPU. */
int ctrlsim_prof_classes(void);
void ctrlsim_prof_enable(int on);
int ctrlsim_prof_collect(double* ms, int64_t* count, double* flops);
int ctrlsim_prof_bytes(double* bytes);
/* ctrlsim_prof_collect + ctrlsim_prof_bytes restricted to the launches on `stream` (on_stream != 0) or on every other stream
 * (on_stream == 0): intervals of kernels that run concurrently on different streams overlap in time. */
int ctrlsim_prof_collect_stream(hipStream_t stream, int on_stream, double* ms, int64_t* count, double* flops, double* bytes);
/* Kernel-level rows inside the two MFMA classes, same restriction by stream.  Row 2 * kind + few: kind 0 other, 1 Linear with K/V
 * image epilogue (QKV / memory K-V projections), 2 Linear + residual + LayerNorm, 3 plain Linear, 4 fused feed-forward block,
 * 5 causal self-attention, 6 key-padded (scene / cross) attention; few = 1 for the few-row launches (last decoder layer on the
 * queried rows, second pass, K/V-cached steps).  Arrays hold ctrlsim_prof_subclasses() entries. */
int ctrlsim_prof_subclasses(void);
int ctrlsim_prof_collect_sub(hipStream_t stream, int on_stream, double* ms, int64_t* count, double* flops, double* bytes);

/* Per-size-class accounting of the causal self-attention launches over the token rows (profiling runs only: one atomic per workgroup).
 * enable != 0: clear and start; enable == 0: stop and (host_out != NULL, 64 x uint64, synchronises) copy out — host_out[2 s] = shader
 * cycles of the workgroups of the classes with s context slots (kernel entry to last store, summed), host_out[2 s + 1] = workgroups. */
int ctrlsim_attn_class_prof(int enable, unsigned long long* host_out);

/* Runtime options (csrc/common.h: OPT_*).  Keys 0 / 1 = attention / GEMM path of the forward: value 0 = f32-input MFMA
 * (v_mfma_f32_32x32x2_f32), 1 = split-operand 16-bit MFMA with fp32-class accuracy (default).  Key 2 = tile shape of the tiled
 * split-operand GEMM (0 auto; tuning).  Key 3 = fused feed-forward block (default 1).  Key 4 = operand split (1 two fp16 planes,
 * 0 three bf16 planes; per engine through ctrlsim_bind).  These are the PROCESS DEFAULTS; an engine overrides them with its own table (ctrlsim_bind_options).  Key 5 = reserved (rounds 2-3: a matrix-pipe variant of the map-encoder pooling, removed).
 * Key 3 value 2 (default) = also the attention out-projection + residual + LayerNorm in front of a feed-forward block as its leading product
 * (ctrlsim_ffn_fused_pre; two-plane scheme); 3 = and a decoder layer's self-attention out-projection + LayerNorm with the cross-attention
 * query projection behind it as one kernel (ctrlsim_outproj_ln_q; measured slightly slower than 2 in the rollout, kept selectable);
 * 1 = the feed-forward block alone; 0 = separate Linear kernels.
 * Key 6 = weight-stationary kernel for the Linear(256 -> 256 G) shapes, bit mask: 1 = launches of at least two 32-row blocks per
 * compute unit, 2 = smaller launches, 4 = the in_proj Linears with K / V-image epilogue, 8 = those through the ROW-stationary kernel
 * (rows in registers, 32-column weight blocks streamed through LDS, every activation row read once; needs the block images of
 * pack.py:row_blocks in the packed weights) inside the forward, 16 = the tall plain 256 -> 256 Linears through it as well (measured
 * slower: off) (default 15; 0 = tiled kernel everywhere).
 * Key 7 = causal self-attention over the token rows takes its visibility masks from the per-class table (default 1; 0 = built per query).
 * Key 8 = the last decoder layer of a rollout pass projects keys / values of every token and queries of the queried tokens only (default 1;
 * 0 = the whole in_proj for every token, then a gather).
 * Key 9 = attention launches with at most 96 queries per context (second pass, last layer on the queried rows, K/V-cached steps) through
 * the streaming form of the kernel: one wave per (context, head, 32 queries), K / V fragments read straight from the tile images
 * (default 1; 0 = the LDS-staged 128-query form for every launch). */
int ctrlsim_set_option(int key, int value);
/* Operand split compiled into the library (csrc/split.h): 1 = two fp16 planes / three products (weights pre-scaled by 2^8), 0 = three
 * bf16 planes / six products.  ctrlsim_amd/pack.py packs weight planes and sizes the K/V images accordingly. */
int ctrlsim_split_scheme(void);
/* Guard events since the last reset.  The guard is a PAIR of device int32 words: [0] non-finite events of the model — sampling races
 * (ctrlsim_sample_rtg / _action) that no finite score won, and rows whose LayerNorm variance was not finite in any fused LayerNorm of the
 * forward (the ReLUs of the MLP heads map NaN to 0, so the logits alone do not show an overflow of the fp16 range of the two-plane operand
 * split) — and [1] simulator events: contacts of an island beyond the solver's table (ctrlsim_sim_step).  Callers check the count (>= 0;
 * synchronises the device) and fail or repeat with ctrlsim_set_option(4, 0).  reset != 0 clears it.  The library's own pair is allocated at
 * first use on the current device; it receives the events of launches made while NO caller-owned pair is bound (ctrlsim_bind below), and
 * this function always reads the library's own pair, never a bound one.  Return value: min(non-finite events, 65535) +
 * 65536 * min(simulator events, 32767) — count % 65536 and count / 65536 tell them apart, and neither kind can carry into the other
 * (the device counts them in separate words). */
int ctrlsim_nonfinite_count(int reset);
/* Per-caller state instead of the process-wide defaults: split_scheme (0 / 1; -1 = leave as is) selects the operand split the
 * caller's weight planes, K/V images and workspace were built for, guard_counter points at TWO device int32 the CALLER owns (zeroed and
 * read by the caller: the library neither allocates nor synchronises for them) that receive every guard event until the next bind —
 * [0] the non-finite events above, [1] the simulator events.  NULL = the library's own pair behind ctrlsim_nonfinite_count.  An engine
 * re-asserts its state at the top of every run, so several engines (planner and adversary policies, a second model) can take turns in one
 * process; still one host thread at a time.  Replaces nothing in the reference: its per-process analogue is nocturne's global Box2D world
 * (physics/Singletons.cpp:5-25). */
int ctrlsim_bind(int split_scheme, int* guard_counter);
/* Before the caller frees a bound pair: un-binds it if it is the bound one (no-op otherwise). */
int ctrlsim_unbind(const int* guard_counter);
/* The pair bound at the moment (NULL = the library's own): a caller that binds its own for ONE call (Simulation.step) restores this one. */
int* ctrlsim_bound_guard(void);
/* Per-engine option table (round 5): values = ctrlsim_option_count() ints in host memory (copied); an entry >= 0 overrides the process
 * default of ctrlsim_set_option for every launch until the next ctrlsim_bind_options, -1 inherits it; NULL = the process defaults.
 * Re-asserted by an engine at the top of every run like ctrlsim_bind: two engines with different kernel options take turns in one process.
 * The operand split (key 4) is the exception: it belongs to ctrlsim_bind, with the weight planes / K/V images / workspace laid out for it;
 * a table's entry for key 4 is ignored.  ctrlsim_get_option = the value launches would use now. */
int ctrlsim_option_count(void);
int ctrlsim_bind_options(const int* values);
int ctrlsim_get_option(int key);

const char* ctrlsim_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CTRLSIM_H */
