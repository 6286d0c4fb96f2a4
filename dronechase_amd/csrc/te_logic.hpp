// te_logic.hpp — per-environment logic of env.step() after the physics: entity operations, offsets,
// navigators, engagement, reward, termination, waves, resets and the observation.  One lane = one env.
//
// Every function is templated on a *view* of the state planes:
//   GView  planes in global memory (reset / observe / set_state fix-up kernels);
//   SView  the engage/observe kernel: the words the logic READS are staged in LDS by one round of
//          independent coalesced loads of the whole block, so the serial per-env logic never waits on a
//          global load; writes go through to global memory (and to the LDS copy when staged).
//
// Reference citations are file:line under the reference's src/ tree.
#pragma once
#include "te_device.hpp"

namespace te {

constexpr int kEPB = 64;   // envs per engage/observe block = one wavefront of logic lanes
constexpr int kMaxD = 32;     // drones per env the LDS kernels (engage_observe_kernel, observe_kernel, the ally kernels) serve: 32-bit slot masks
constexpr int kMaxD64 = 64;   // ... and what the state layout, the sub-step kernel, engage_kernel<7, 30> and stacked_kernel serve (Level5DumbMultiObs: 7 + 30)
enum Family { FAM_LEVEL4 = 0, FAM_STAGE01 = 1, FAM_STAGE02 = 2 };

// ------------------------------------------------------------------------------------------------
// views
// ------------------------------------------------------------------------------------------------
struct GView {
  uint32_t* d; uint32_t* e; int D; int Npad; int env; int P;
  TE_DEV size_t ix(int w, int s) const { return ((size_t)w * D + s) * Npad + env; }
  TE_DEV float gf(int w, int s) const { return __uint_as_float(d[ix(w, s)]); }
  TE_DEV int gi(int w, int s) const { return (int)d[ix(w, s)]; }
  TE_DEV void sf(int w, int s, float v) const { d[ix(w, s)] = __float_as_uint(v); }
  TE_DEV void si(int w, int s, int v) const { d[ix(w, s)] = (uint32_t)v; }
  TE_DEV float egf(int w) const { return __uint_as_float(e[(size_t)w * Npad + env]); }
  TE_DEV int egi(int w) const { return (int)e[(size_t)w * Npad + env]; }
  TE_DEV void esf(int w, float v) const { e[(size_t)w * Npad + env] = __float_as_uint(v); }
  TE_DEV void esi(int w, int v) const { e[(size_t)w * Npad + env] = (uint32_t)v; }
};

// LDS rows (kEPB words each) of the engage/observe kernel
struct Rows {
  int D, P;
  __host__ __device__ __forceinline__ int I() const { return D - P; }
  __host__ __device__ __forceinline__ int obs_pos() const { return 0; }                  // 3*D : word-major, slot-minor
  __host__ __device__ __forceinline__ int armed() const { return 3 * D; }                // D   : ARMED
  __host__ __device__ __forceinline__ int munition() const { return 4 * D; }             // P   : MUNITION of the pursuers (nobody reads an invader's gun)
  __host__ __device__ __forceinline__ int last_fired() const { return 4 * D + P; }       // P   : LAST_FIRED of the pursuers
  __host__ __device__ __forceinline__ int agent() const { return 4 * D + 2 * P; }        // 9   : OBS_EULER, OBS_VEL, OBS_RATE of slot 0
  __host__ __device__ __forceinline__ int env() const { return agent() + 9; }            // TE_ENV_WORDS
  __host__ __device__ __forceinline__ int staged() const { return env() + TE_ENV_WORDS; }  // rows loaded from global memory
  __host__ __device__ __forceinline__ int rinv() const { return staged(); }              // 9   : inverse attitude of the agent (row-major)
  __host__ __device__ __forceinline__ int dpi() const { return rinv() + 9; }             // P*I : |obs_pos(p) - obs_pos(j)|
  __host__ __device__ __forceinline__ int zone() const { return dpi() + P * I(); }       // 1   : bit s = outside dome, bit 16+.. unused; see zone bits
  __host__ __device__ __forceinline__ int origin() const { return zone() + 1; }          // 1   : bit s = |obs_pos(s)| < origin_range
  __host__ __device__ __forceinline__ int lcell() const { return origin() + 1; }         // D   : LIDAR cell of drone j seen from the agent
  __host__ __device__ __forceinline__ int lrhat() const { return lcell() + D; }          // D   : normalised range of drone j
  __host__ __device__ __forceinline__ int hitmask() const { return lrhat() + D; }        // 1   : bit j = drone j owns its cell in this step's sphere
  __host__ __device__ __forceinline__ int done() const { return hitmask() + 1; }         // 1   : env auto-reset this step
  __host__ __device__ __forceinline__ int prevalid() const { return done() + 1; }        // 1   : distance / zone rows still valid after the logic
  __host__ __device__ __forceinline__ int task() const { return prevalid() + 1; }        // 1   : level4 spawn work left to the block: round | reset << 8
  __host__ __device__ __forceinline__ int smask() const { return task() + 1; }           // 1   : bit s = drone s armed when the block was staged
  __host__ __device__ __forceinline__ int anow() const { return smask() + 1; }           // 1   : bit s = drone s armed after the engagement (observation time)
  __host__ __device__ __forceinline__ int sstep() const { return smask() + 2; }          // 1   : RL step of this observation (the env record may be reset after it)
  __host__ __device__ __forceinline__ int sepis() const { return smask() + 3; }          // 1   : episode of this observation
  __host__ __device__ __forceinline__ int total() const { return sepis() + 1; }
};
__host__ __device__ inline int lds_rows(int D, int P) { return 6 * D + 2 * P + P * (D - P) + 9 + TE_ENV_WORDS + 9 + 10; }

struct SView {
  GView g; uint32_t* sm; int lane; Rows r;
  int D, P, env;
  mutable bool pre_valid;  // the precomputed distance / zone rows still describe the current positions
  TE_DEV int at(int row) const { return row * kEPB + lane; }
  // LDS address of a staged drone word, or -1
  TE_DEV int dmap(int w, int s) const {
    if (w >= TE_D_OBS_POS && w < TE_D_OBS_POS + 3) return at(r.obs_pos() + (w - TE_D_OBS_POS) * D + s);
    if (w == TE_D_ARMED) return at(r.armed() + s);
    if (w == TE_D_MUNITION && s < P) return at(r.munition() + s);
    if (w == TE_D_LAST_FIRED && s < P) return at(r.last_fired() + s);
    if (s == 0 && w >= TE_D_OBS_EULER && w < TE_D_OBS_EULER + 9) return at(r.agent() + (w - TE_D_OBS_EULER));
    return -1;
  }
  TE_DEV float gf(int w, int s) const { int a = dmap(w, s); return a >= 0 ? __uint_as_float(sm[a]) : g.gf(w, s); }
  TE_DEV int gi(int w, int s) const { int a = dmap(w, s); return a >= 0 ? (int)sm[a] : g.gi(w, s); }
  TE_DEV void sf(int w, int s, float v) const {
    int a = dmap(w, s);
    if (a >= 0) { sm[a] = __float_as_uint(v); if (w >= TE_D_OBS_POS && w < TE_D_OBS_POS + 3) pre_valid = false; }
    g.sf(w, s, v);
  }
  TE_DEV void si(int w, int s, int v) const { int a = dmap(w, s); if (a >= 0) sm[a] = (uint32_t)v; g.si(w, s, v); }
  TE_DEV float egf(int w) const { return __uint_as_float(sm[at(r.env() + w)]); }
  TE_DEV int egi(int w) const { return (int)sm[at(r.env() + w)]; }
  TE_DEV void esf(int w, float v) const { sm[at(r.env() + w)] = __float_as_uint(v); g.esf(w, v); }
  TE_DEV void esi(int w, int v) const { sm[at(r.env() + w)] = (uint32_t)v; g.esi(w, v); }
};

// ------------------------------------------------------------------------------------------------
// entity operations (quadcopter.py:433-478, gun.py)
// ------------------------------------------------------------------------------------------------
template <class V> TE_DEV V3 obs_pos(const V& v, int s) { return V3{v.gf(TE_D_OBS_POS, s), v.gf(TE_D_OBS_POS + 1, s), v.gf(TE_D_OBS_POS + 2, s)}; }
TE_DEV float dist(V3 a, V3 b) { return norm(sub(a, b)); }
// pursuer p <-> invader j distance, |obs_pos(s)| > dome, |obs_pos(s)| < origin_range
// (OffsetHandler.compute_distances_and_directions, offsets_handler.py:150-164; :341-391)
TE_DEV float pi_dist(const GView& v, int p, int j) { return dist(obs_pos(v, p), obs_pos(v, j)); }
TE_DEV bool outside_dome(const te_config& c, const GView& v, int s) { return norm(obs_pos(v, s)) > c.dome_radius; }
TE_DEV bool in_origin(const te_config& c, const GView& v, int s) { return norm(obs_pos(v, s)) < c.origin_range; }
TE_DEV float pi_dist(const SView& v, int p, int j) {
  if (v.pre_valid) return __uint_as_float(v.sm[v.at(v.r.dpi() + p * (v.D - v.P) + (j - v.P))]);
  return dist(obs_pos(v, p), obs_pos(v, j));
}
TE_DEV bool outside_dome(const te_config& c, const SView& v, int s) {
  if (v.pre_valid) return (v.sm[v.at(v.r.zone())] >> s) & 1u;
  return norm(obs_pos(v, s)) > c.dome_radius;
}
TE_DEV bool in_origin(const te_config& c, const SView& v, int s) {
  if (v.pre_valid) return (v.sm[v.at(v.r.origin())] >> s) & 1u;
  return norm(obs_pos(v, s)) < c.origin_range;
}

// Quadcopter.disarm (quadcopter.py:461-478): static body, velocities zeroed, motors/body/set-point/pwm
// reset.  PID memories and the last IMU read stay.
template <class V> TE_DEV void disarm(const V& v, int s) {
  v.si(TE_D_ARMED, s, 0);
#pragma unroll
  for (int k = 0; k < 3; ++k) { v.sf(TE_D_VEL + k, s, 0.0f); v.sf(TE_D_OMEGA + k, s, 0.0f); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { v.sf(TE_D_THROTTLE + k, s, 0.0f); v.sf(TE_D_SETPOINT + k, s, 0.0f); }
}
// Quadcopter.replace (quadcopter.py:433-439): teleport, identity attitude, zero base velocity
template <class V> TE_DEV void replace_planes(const V& v, int s, V3 p) {
  v.sf(TE_D_POS, s, p.x); v.sf(TE_D_POS + 1, s, p.y); v.sf(TE_D_POS + 2, s, p.z);
  v.sf(TE_D_FORMATION, s, p.x); v.sf(TE_D_FORMATION + 1, s, p.y); v.sf(TE_D_FORMATION + 2, s, p.z);
  v.sf(TE_D_QUAT, s, 0.0f); v.sf(TE_D_QUAT + 1, s, 0.0f); v.sf(TE_D_QUAT + 2, s, 0.0f); v.sf(TE_D_QUAT + 3, s, 1.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) { v.sf(TE_D_VEL + k, s, 0.0f); v.sf(TE_D_OMEGA + k, s, 0.0f); }
}
// IMU (imu.py:27-41) of a drone that has just been teleported: identity attitude, at rest
template <class V> TE_DEV void observe_at_rest(const V& v, int s, V3 p) {
  v.sf(TE_D_OBS_POS, s, p.x); v.sf(TE_D_OBS_POS + 1, s, p.y); v.sf(TE_D_OBS_POS + 2, s, p.z);
#pragma unroll
  for (int k = 0; k < 3; ++k) { v.sf(TE_D_OBS_EULER + k, s, 0.0f); v.sf(TE_D_OBS_VEL + k, s, 0.0f); v.sf(TE_D_OBS_RATE + k, s, 0.0f); }
}
TE_DEV int max_munition_of(const te_config& c, int slot) {
  if (slot >= c.n_pursuers) return 10;  // Gun default (gun.py:11)
  if (c.task == TE_TASK_STAGE02) return slot == 0 ? c.munition : 10;  // stages.py:118
  return c.munition;
}
// disarm -> replace -> arm of a drone that ends up armed at p (Task.setup_round,
// exp03_vFinal_task.py:180-196): arm() reads the IMU and resets the gun (quadcopter.py:445-459)
template <class V> TE_DEV void respawn_armed(const te_config& c, const V& v, int s, V3 p) {
  disarm(v, s);
  replace_planes(v, s, p);
  observe_at_rest(v, s, p);
  v.si(TE_D_ARMED, s, 1);
  v.si(TE_D_MUNITION, s, max_munition_of(c, s));
  v.si(TE_D_LAST_FIRED, s, -c.cooldown_steps);
}
// gun.py:56-75
TE_DEV bool gun_available(const te_config& c, int munition, int last_fired, int step) {
  return munition <= 0 || c.cooldown_steps <= step - last_fired;
}
// gun.py:101-113
TE_DEV void gun_state(const te_config& c, int munition, int last_fired, int step, int max_mun, float g[3]) {
  float wait = fmaxf((float)c.cooldown_steps - (float)(step - last_fired), 0.0f);
  g[0] = (float)munition / (float)(max_mun > 0 ? max_mun : 1);
  g[1] = wait / (float)c.cooldown_steps;
  g[2] = gun_available(c, munition, last_fired, step) ? 1.0f : 0.0f;
}
template <class V> TE_DEV uint32_t armed_mask(const V& v) {
  uint32_t m = 0;
  for (int s = 0; s < v.D && s < 32; ++s) m |= (v.gi(TE_D_ARMED, s) ? 1u : 0u) << s;   // (slots 32.. : SnapRows::armed_hi, engage_kernel's 64-bit masks)
  return m;
}

// spawn samplers ---------------------------------------------------------------------------------
// Task.generate_positions (exp03_vFinal_task.py:584-608)
// sin and cos of x in [0, pi] (the two angles of a spawn position): Cody-Waite reduction to |r| <= pi/4 around 0, pi/2 or pi and the
// Cephes sinf / cosf minimax polynomials, < 1 ulp of the result's scale.  libm's sincosf carries its large-argument reduction along:
// a respawning wave of the engage kernel paid ~0.3 us per call on its critical path (it is the kernel's slowest wave).
TE_DEV void sincos_0_pi(float x, float& s, float& c) {
  TE_EXACT   // (inlined into several kernels whose spawns are compared bitwise: te_device.hpp "exact arithmetic")
  const int q = x < 0.7853981633974483f ? 0 : (x < 2.356194490192345f ? 1 : 2);   // nearest multiple of pi/2
  const float y = (float)q * 2.0f;                                              // in units of pi/4
  const float r = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
  const float z = r * r;
  const float sr = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
  const float cr = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
  s = q == 0 ? sr : (q == 1 ? cr : -sr);
  c = q == 0 ? cr : (q == 1 ? -sr : -cr);
}
TE_DEV V3 level4_position(const te_config& c, float r, float u_theta, float u_phi) {
  TE_EXACT
  float theta = u_theta * kPi;
  float lower = fminf(c.born_min_z, r);
  float min_phi = 0.5f * kPi - x_asin(lower * rcp(r));   // acos on the polynomial asin of the sub-step loop (1e-7 abs)
  float phi = (r >= c.born_min_z) ? min_phi + u_phi * (0.5f * kPi - min_phi) : u_phi * (0.5f * kPi);
  float sph, cph, sth, cth;
  sincos_0_pi(phi, sph, cph); sincos_0_pi(theta, sth, cth);
  return V3{r * sph * cth, r * sph * sth, r * cph};
}
// L3Stage1.generate_positions (level3/components/stages.py:360-376)
TE_DEV V3 stage02_position(float r, float r_max, float u_r, float u_theta, float u_phi) {
  TE_EXACT   // (inlined into engage_stage02_kernel and engage_slots_stage02_kernel, whose respawns are compared bitwise)
  if (r > r_max) r_max = r;
  float radius = r + u_r * (r_max - r);
  float theta = u_theta * 2.0f * kPi, phi = u_phi * kPi * 0.5f;
  float sph, cph, sth, cth;
  sincosf(phi, &sph, &cph); sincosf(theta, &sth, &cth);
  return V3{radius * sph * cth, radius * sph * sth, radius * cph};
}

// invaders armed in round r: `round` of them in the exp tasks (exp03_vFinal_task.py:180-196); min((r - 1) * per_round + initial, max) in
// Level5DumbMultiObjectTask (level5_dumb_multiobject_task.py:173-184)
TE_DEV int invaders_in_round(const te_config& c, int round) { return min((round - 1) * c.invaders_per_round + c.initial_invaders, c.n_invaders); }
// Evaluation_Task / Level5DumbMultiObjectTask: pursuer 0 obeys the behaviour tree like the other wingmen
__host__ __device__ __forceinline__ bool all_scripted(const te_config& c) { return c.evaluation != 0 || c.agent_scripted != 0; }
template <class V> TE_DEV void set_snap_mask(const V& v, uint64_t m) { v.esi(TE_E_SNAP_MASK, (int)(uint32_t)m); v.esi(TE_E_SNAP_MASK_HI, (int)(uint32_t)(m >> 32)); }
template <class V> TE_DEV uint64_t get_snap_mask(const V& v) { return (uint64_t)(uint32_t)v.egi(TE_E_SNAP_MASK) | ((uint64_t)(uint32_t)v.egi(TE_E_SNAP_MASK_HI) << 32); }
// ------------------------------------------------------------------------------------------------
// offsets over the snapshot mask (level4/components/entities_management/offsets_handler.py)
// ------------------------------------------------------------------------------------------------
// closest invader (over mask) to pursuer p, and its distance (identify_closest_invader, :256-281)
template <class V> TE_DEV int closest_invader(const V& v, uint64_t mask, int p, float& dmin) {
  int best = -1; dmin = 0.0f;
  for (int j = v.P; j < v.D; ++j) {
    if (!((mask >> j) & 1u)) continue;
    float d = pi_dist(v, p, j);
    if (best < 0 || d < dmin) { best = j; dmin = d; }
  }
  return best;
}
// closest pursuer (over mask) to invader j (identify_closest_pursuer, :228-254)
template <class V> TE_DEV int closest_pursuer(const V& v, uint64_t mask, int j) {
  int best = -1; float bd = 0.0f;
  for (int p = 0; p < v.P; ++p) {
    if (!((mask >> p) & 1u)) continue;
    float d = pi_dist(v, p, j);
    if (best < 0 || d < bd) { best = p; bd = d; }
  }
  return best;
}
// closest other pursuer to pursuer p (identify_closest_ally, :167-190)
template <class V> TE_DEV int closest_ally(const V& v, uint64_t mask, int p) {
  if (!((mask >> p) & 1u) || __popc((uint32_t)mask & ((1u << v.P) - 1u)) <= 1) return -1;
  V3 me = obs_pos(v, p);
  int best = -1; float bd = 0.0f;
  for (int a = 0; a < v.P; ++a) {
    if (a == p || !((mask >> a) & 1u)) continue;
    float d = dist(obs_pos(v, a), me);
    if (best < 0 || d < bd) { best = a; bd = d; }
  }
  return best;
}

// ------------------------------------------------------------------------------------------------
// scripted commands of the NEXT step (Task.on_step_start, exp03_vFinal_task.py:232-244,276-283).
// They depend only on the state at the end of this step.  The allies' are prepared here (engage/observe kernel)
// and read back by the sub-step kernel; each invader works its own out at the top of the sub-step kernel.
// ------------------------------------------------------------------------------------------------
// cmd_toward + Quadcopter.convert_command_to_setpoint (te_device.hpp: command_to_velocity) for the behaviour tree's command, in exact
// arithmetic (te_device.hpp): prepare_slot (after te_reset / te_set_state) and the engage kernels (during a rollout) must write the same bits
TE_DEV void x_cmd_toward(V3 from, V3 to, float speed, float out[3]) {
  TE_EXACT
  const float dx = to.x - from.x, dy = to.y - from.y, dz = to.z - from.z;
  const float n = sqrtf(xfma(dz, dz, xfma(dy, dy, dx * dx)));
  const float inv = n > 0.0f ? 1.0f / n : 1.0f;  // zero vector stays zero (…air_combat_only.py:191-195)
  const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
  const float m = sqrtf(xfma(uz, uz, xfma(uy, uy, ux * ux)));
  const float inv2 = 1.0f / (m > 0.0f ? m : 1.0f);
  out[0] = speed * (ux * inv2); out[1] = speed * (uy * inv2); out[2] = speed * (uz * inv2);
}
TE_DEV void cmd_toward(V3 from, V3 to, float speed, float out[3]) { x_cmd_toward(from, to, speed, out); }   // (one arithmetic for the navigators of the sub-step kernel and the engage kernels)
// GeometryUtils.is_point_inside_cone (geometry_utils.py:6-29)
TE_DEV bool inside_cone(V3 p, V3 apex, V3 base, float degrees) {
  V3 ab = sub(base, apex), ap = sub(p, apex);
  float nab = norm(ab), nap = norm(ap);
  if (nap > nab) return false;
  float cosang = (ap.x * ab.x + ap.y * ab.y + ap.z * ab.z) / (nap * nab);
  return acosf(cosang) * (180.0f / kPi) <= 0.5f * degrees;
}
template <class V> TE_DEV bool building_path_clear(const te_config& c, const V& v, uint32_t mask, V3 me, float degrees) {
  if (!c.kamikaze_cone_check) return false;  // …air_combat_only.py:83-96: constant False
  V3 b{c.building_position[0], c.building_position[1], c.building_position[2]};
  for (int p = 0; p < c.n_pursuers; ++p)
    if (((mask >> p) & 1u) && inside_cone(obs_pos(v, p), me, b, degrees)) return false;
  return true;
}
// KamikazeNavigator.update of invader `s` (…air_combat_only.py:68-78): the transition is registered, the OLD
// state executes.  Pure: reads the snapshot mask S, the invader's FSM state and IMU position `me`, and the
// pursuers' IMU positions through `v`; returns the velocity command and the next FSM state.  Evaluated by the
// invader's own wave at the top of the sub-step kernel (Task.on_step_start, exp03_vFinal_task.py:276-283).
template <class V> TE_DEV int kamikaze_update(const te_config& c, const V& v, uint32_t S, int state, V3 me, float out[3]) {
  const int Pn = c.n_pursuers;
  const uint32_t pursuer_bits = S & ((1u << Pn) - 1u);
  int next = state;
  if (state == TE_NAV_WAIT) {
    if (building_path_clear(c, v, S, me, 60.0f)) next = TE_NAV_COLLIDE_BUILDING;
    else if (pursuer_bits) next = TE_NAV_COLLIDE_WINGMAN;
    out[0] = out[1] = out[2] = 0.0f;  // hover (:163)
  } else if (state == TE_NAV_COLLIDE_WINGMAN) {
    if (!pursuer_bits) next = TE_NAV_COLLIDE_BUILDING;
    // identify_closest_pursuer (offsets_handler.py:228-254) from the IMU positions
    int best = -1; float bd = 0.0f; V3 target{0, 0, 0};
    for (int p = 0; p < Pn; ++p) {
      if (!((S >> p) & 1u)) continue;
      const V3 pp = obs_pos(v, p);
      const float d = dist(pp, me);
      if (best < 0 || d < bd) { best = p; bd = d; target = pp; }
    }
    cmd_toward(me, target, c.invader_speed, out);
  } else {
    if (!building_path_clear(c, v, S, me, 45.0f)) next = TE_NAV_COLLIDE_WINGMAN;
    cmd_toward(me, V3{c.building_position[0], c.building_position[1], c.building_position[2]}, c.invader_speed, out);
  }
  return next;
}
// is pursuer s flown by the caller (te_set_wingman_actions) instead of a scripted driver?  exp05's ally, or a pursuer of
// the Evaluation_Task driver mask (cfg.evaluation bits 8..)
TE_DEV bool driven_externally(const te_config& c, int s) {
  return (c.ally_policy == TE_ALLY_EXTERNAL && s == 1) || (((uint32_t)c.evaluation >> (8 + s)) & 1u) != 0u;
}
// command of ONE scripted ally `s` for the NEXT step (LoyalWingmanBehaviorTree.update), stored in the TE_X_CMD
// planes by the engage/observe kernel, where the pursuer-invader distances are at hand in LDS
template <class V> TE_DEV void prepare_slot(const te_config& c, const V& v, int s) {
  const int Pn = c.n_pursuers;
  // Evaluation_Task.drive_lw (evaluation_task.py:257-275) flies EVERY armed pursuer, pursuer 0 included
  if ((s == 0 && !all_scripted(c)) || s >= Pn || !v.gi(TE_D_ARMED, s)) return;
  const uint64_t S = get_snap_mask(v);
  // drive_loyalwingmen: get_armed_pursuers()[1:] (exp03_vFinal_task.py:238-244): with the agent dead the
  // first armed ally is the one that is skipped
  if (!all_scripted(c) && !v.gi(TE_D_ARMED, 0)) {
    int first = -1;
    for (int a = 1; a < Pn; ++a) if (v.gi(TE_D_ARMED, a)) { first = a; break; }
    if (s == first) return;
  }
  float out[3] = {0.0f, 0.0f, 0.0f};
  const bool ext = driven_externally(c, s);
  if (!ext && c.ally_policy == TE_ALLY_BT) {  // LoyalWingmanBehaviorTree (loyalwingman_navigator.py:238-352)
    V3 me = obs_pos(v, s);
    if (gun_available(c, v.gi(TE_D_MUNITION, s), v.gi(TE_D_LAST_FIRED, s), v.egi(TE_E_STEP))) {
      float dm;
      int t = ((S >> s) & 1u) ? closest_invader(v, S, s, dm) : -1;
      V3 target = t >= 0 ? obs_pos(v, t) : V3{0, 0, 0};
      x_cmd_toward(me, target, c.ally_speed, out);
    } else {
      x_cmd_toward(me, V3{v.gf(TE_D_FORMATION, s), v.gf(TE_D_FORMATION + 1, s), v.gf(TE_D_FORMATION + 2, s)}, c.ally_speed, out);
    }
  } else if (ext || c.ally_policy != TE_ALLY_FROZEN) {  // (frozen: exp04_vFinal_task.py:240-242: drive([0,0,0,1]))
    // nobody, or the caller's policy (te_set_ally_actions, exp05): the set-point persists
    out[0] = v.gf(TE_D_SETPOINT + 0, s); out[1] = v.gf(TE_D_SETPOINT + 1, s); out[2] = v.gf(TE_D_SETPOINT + 3, s);
  }
  v.sf(TE_X_CMD + 0, s, out[0]); v.sf(TE_X_CMD + 1, s, out[1]); v.sf(TE_X_CMD + 2, s, out[2]);
}
// the pursuers' positions the invaders will steer at during the next sub-step launch (TE_X_REF)
template <class V> TE_DEV void publish_pursuer_ref(const V& v, int s) {
  v.sf(TE_X_REF + 0, s, v.gf(TE_D_OBS_POS + 0, s)); v.sf(TE_X_REF + 1, s, v.gf(TE_D_OBS_POS + 1, s));
  v.sf(TE_X_REF + 2, s, v.gf(TE_D_OBS_POS + 2, s));
}
template <class V> TE_DEV void prepare_level4_commands(const te_config& c, const V& v) {
  for (int s = 0; s < c.n_pursuers; ++s) publish_pursuer_ref(v, s);
  for (int s = all_scripted(c) ? 0 : 1; s < c.n_pursuers; ++s) prepare_slot(c, v, s);
}

// ------------------------------------------------------------------------------------------------
// resets
// ------------------------------------------------------------------------------------------------
// Task.setup_round (exp03_vFinal_task.py:180-196) and Task.on_reset (:255-274), as ONE function per drone slot so
// that the engage/observe kernel can hand the slots of a respawning env to the threads of its block (spawn
// phase) instead of leaving all D of them to the env's own logic lane.  The draws are keyed on (slot, episode,
// round), so the order in which slots are processed does not matter.
//   invader slot: disarm; the first min(round, n_invaders) slots respawn armed on the born sphere
//   pursuer slot: untouched by a new round; on reset, disarm + respawn armed on the pursuer sphere
//   every slot  : navigator reset (OffsetHandler snapshot refresh + navigators reset())
template <class V> TE_DEV void level4_spawn_slot(const te_config& c, const V& v, int s, int round, uint32_t episode, bool reset) {
  const int Pn = c.n_pursuers;
  if (s >= Pn) {
    disarm(v, s);
    const int i = s - Pn;
    if (i < invaders_in_round(c, round)) {
      U4 r = env_rng(c, v.env, RNG_SPAWN_INVADER, (uint32_t)s, 0, episode, (uint32_t)round);
      respawn_armed(c, v, s, level4_position(c, c.born_radius, u01(r.x), u01(r.y)));
    }
  } else if (reset) {
    U4 r = env_rng(c, v.env, RNG_SPAWN_PURSUER, (uint32_t)s, 0, episode, 0);
    respawn_armed(c, v, s, level4_position(c, c.pursuer_spawn_radius, u01(r.x), u01(r.y)));
    if (driven_externally(c, s)) {  // Exp05_vFinal_Task.init_globals: last_action = zeros (exp05_vFinal_task.py:139)
#pragma unroll
      for (int k = 0; k < 4; ++k) v.sf(TE_D_ALLY_ACTION + k, s, 0.0f);
    }
  }
  // navigators reset(); a pursuer's word counts its kills of the episode under cfg.evaluation and survives the waves
  if (s >= Pn || reset) v.si(TE_D_NAV_STATE, s, TE_NAV_WAIT);
}
// armed mask once every slot has been through level4_spawn_slot
template <class V> TE_DEV uint64_t level4_mask_after_spawn(const te_config& c, const V& v, int round, bool reset) {
  const int Pn = c.n_pursuers;
  uint64_t m = 0;
  for (int p = 0; p < Pn; ++p) m |= (uint64_t)((reset || v.gi(TE_D_ARMED, p)) ? 1u : 0u) << p;
  const int n = invaders_in_round(c, round);
  return m | ((((uint64_t)1 << n) - 1u) << Pn);
}
template <class V> TE_DEV void level4_setup_round(const te_config& c, const V& v, int round, uint32_t episode) {
  for (int s = 0; s < v.D; ++s) level4_spawn_slot(c, v, s, round, episode, false);
  set_snap_mask(v, level4_mask_after_spawn(c, v, round, false));
}
// the env-record half of Env.reset -> Task.on_reset (exp03_vFinal_environment.py:128-146)
template <class V> TE_DEV uint32_t level4_reset_record(const te_config& c, const V& v) {
  const uint32_t episode = (uint32_t)(v.egi(TE_E_EPISODE) + 1);
  v.esi(TE_E_EPISODE, (int)episode);
  v.esi(TE_E_STEP, 0); v.esi(TE_E_MAX_STEP, c.max_step); v.esi(TE_E_ROUND, 1); v.esi(TE_E_INFO_WAVE, 1);
  v.esi(TE_E_AGENT_KILLS, 0); v.esi(TE_E_ALLIES_KILLS, 0); v.esi(TE_E_DEADS, 0);
  if (c.reward_model != TE_REWARD_L5_C1) v.esf(TE_E_LAST_DIST, c.dome_radius);   // Level5C1FusionTask's last_distance outlives every reset
#pragma unroll
  for (int k = 0; k < 4; ++k) v.esf(TE_E_LAST_ACTION + k, 0.0f);
  set_snap_mask(v, level4_mask_after_spawn(c, v, 1, true));
  return episode;
}
template <class V> TE_DEV void level4_reset_env(const te_config& c, const V& v, bool prepare = true) {
  const uint32_t episode = level4_reset_record(c, v);
  for (int s = 0; s < v.D; ++s) level4_spawn_slot(c, v, s, 1, episode, true);
  if (prepare) prepare_level4_commands(c, v);
}

// stage02 ----------------------------------------------------------------------------------------
template <class V> TE_DEV float stage02_agent_min_distance(const te_config& c, const V& v, uint32_t S) {
  int first = -1;
  for (int p = 0; p < c.n_pursuers; ++p) if ((S >> p) & 1u) { first = p; break; }
  if (first < 0) return 0.0f;
  float dmin;
  int j = closest_invader(v, S, first, dmin);
  return j >= 0 ? dmin : 0.0f;
}
TE_DEV V3 stage02_invader_position(const te_config& c, int env, int slot, uint32_t episode, uint32_t tag) {
  U4 r = env_rng(c, env, RNG_RESPAWN, (uint32_t)slot, 0, episode, tag);
  return stage02_position(2.0f, 6.0f, u01(r.x), u01(r.y), u01(r.z));  // stages.py:378-384
}
// L3Stage1.on_reset (stages.py:104-131)
template <class V> TE_DEV void stage02_reset_env(const te_config& c, const V& v) {
  const uint32_t episode = (uint32_t)(v.egi(TE_E_EPISODE) + 1);
  v.esi(TE_E_EPISODE, (int)episode);
  v.esi(TE_E_STEP, 0); v.esi(TE_E_MAX_STEP, c.max_step); v.esi(TE_E_ROUND, 0);
  v.esi(TE_E_AGENT_KILLS, 0); v.esi(TE_E_ALLIES_KILLS, 0); v.esi(TE_E_DEADS, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) v.esf(TE_E_LAST_ACTION + k, 0.0f);
  for (int j = c.n_pursuers; j < v.D; ++j) respawn_armed(c, v, j, stage02_invader_position(c, v.env, j, episode, 0u));
  for (int p = 0; p < c.n_pursuers; ++p) {
    U4 r = env_rng(c, v.env, RNG_SPAWN_PURSUER, (uint32_t)p, 0, episode, 0);
    respawn_armed(c, v, p, stage02_position(c.pursuer_spawn_radius, 0.0f, u01(r.x), u01(r.y), u01(r.z)));
  }
  uint32_t S = armed_mask(v);
  v.esi(TE_E_SNAP_MASK, (int)S);
  float d0 = stage02_agent_min_distance(c, v, S);
  v.esf(TE_E_PREV_SNAP_MIN, d0); v.esf(TE_E_LAST_DIST, d0);
}

// stage01 ----------------------------------------------------------------------------------------
TE_DEV V3 stage01_cube(const te_config& c, int env, uint32_t purpose, uint32_t slot, uint32_t episode, uint32_t index) {
  U4 r = env_rng(c, env, purpose, slot, 0, episode, index);
  return V3{-1.0f + 2.0f * u01(r.x), -1.0f + 2.0f * u01(r.y), -1.0f + 2.0f * u01(r.z)};
}
// QuadcopterManager.replace_invader (level2/components/quadcopter_manager.py:166-179): teleport, raw mode-7
// set-point, then ONE extra imu/control/physics update whose wrench stays accumulated until the next
// stepSimulation.
template <class V> TE_DEV void stage01_replace_invader(const te_config& c, const V& v, V3 p, uint32_t episode, uint32_t step_index) {
  const int s = 2;
  replace_planes(v, s, p);
  observe_at_rest(v, s, p);
  float sp[4] = {p.x, p.y, 0.0f, p.z};
#pragma unroll
  for (int k = 0; k < 4; ++k) v.sf(TE_D_SETPOINT + k, s, sp[k]);
  Body b;
  b.pos = p; b.q = Q4{0, 0, 0, 1}; b.vel = V3{0, 0, 0}; b.wb = V3{0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) b.thr[k] = v.gf(TE_D_THROTTLE + k, s);
#pragma unroll
  for (int k = 0; k < 3; ++k) { b.av_i[k] = v.gf(TE_D_PID_AV_I + k, s); b.av_e[k] = v.gf(TE_D_PID_AV_E + k, s); }
#pragma unroll
  for (int k = 0; k < 2; ++k) { b.lv_i[k] = v.gf(TE_D_PID_LV_I + k, s); b.lv_e[k] = v.gf(TE_D_PID_LV_E + k, s); }
  b.zv_i = v.gf(TE_D_PID_ZV_I, s); b.zv_e = v.gf(TE_D_PID_ZV_E, s);
  // controller + motors only: the body must not move, so evaluate the wrench by differencing a
  // throw-away sub-step from rest (velocity change * mass / dt = applied force, etc.)
  V3 pf{0, 0, 0}, pt{0, 0, 0};
  if (c.motor_noise) {  // sub-step index 255: Philox call 127, words {z, w}
    const U4 bits = motor_noise_bits(c, v.env, s, episode, step_index, 255);
    substep<true, false, true>(c, derive(c), b, sp, bits.z, bits.w, pf, pt);
  } else substep<true, false, false>(c, derive(c), b, sp, 0u, 0u, pf, pt);
  const float dt = c.physics_dt;
  V3 F{b.vel.x * c.quad.mass / dt, b.vel.y * c.quad.mass / dt, (b.vel.z / dt + c.quad.gravity) * c.quad.mass};
  V3 Tq{b.wb.x * c.quad.inertia[0] / dt, b.wb.y * c.quad.inertia[1] / dt, b.wb.z * c.quad.inertia[2] / dt};
  v.sf(TE_D_PENDING + 0, s, v.gf(TE_D_PENDING + 0, s) + F.x); v.sf(TE_D_PENDING + 1, s, v.gf(TE_D_PENDING + 1, s) + F.y);
  v.sf(TE_D_PENDING + 2, s, v.gf(TE_D_PENDING + 2, s) + F.z); v.sf(TE_D_PENDING + 3, s, v.gf(TE_D_PENDING + 3, s) + Tq.x);
  v.sf(TE_D_PENDING + 4, s, v.gf(TE_D_PENDING + 4, s) + Tq.y); v.sf(TE_D_PENDING + 5, s, v.gf(TE_D_PENDING + 5, s) + Tq.z);
#pragma unroll
  for (int k = 0; k < 4; ++k) v.sf(TE_D_THROTTLE + k, s, b.thr[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { v.sf(TE_D_PID_AV_I + k, s, b.av_i[k]); v.sf(TE_D_PID_AV_E + k, s, b.av_e[k]); }
#pragma unroll
  for (int k = 0; k < 2; ++k) { v.sf(TE_D_PID_LV_I + k, s, b.lv_i[k]); v.sf(TE_D_PID_LV_E + k, s, b.lv_e[k]); }
  v.sf(TE_D_PID_ZV_I, s, b.zv_i); v.sf(TE_D_PID_ZV_E, s, b.zv_e);
}
// PyflytL2EnviromentModifiedV2.reset (pyflyt_level2_environment_modified_v2.py:83-123)
template <class V> TE_DEV void stage01_reset_env(const te_config& c, const V& v) {
  const uint32_t episode = (uint32_t)(v.egi(TE_E_EPISODE) + 1);
  v.esi(TE_E_EPISODE, (int)episode);
  v.esi(TE_E_STEP, 0); v.esi(TE_E_MAX_STEP, c.max_step); v.esi(TE_E_ROUND, 0);
  v.esi(TE_E_AGENT_KILLS, 0); v.esi(TE_E_ALLIES_KILLS, 0); v.esi(TE_E_DEADS, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) v.esf(TE_E_LAST_ACTION + k, 0.0f);
  for (int s = 0; s < v.D; ++s)
    if (!v.gi(TE_D_ARMED, s)) { v.si(TE_D_ARMED, s, 1); v.si(TE_D_MUNITION, s, 0); v.si(TE_D_LAST_FIRED, s, -c.cooldown_steps); }
  V3 pi = stage01_cube(c, v.env, RNG_SPAWN_INVADER, 2, episode, 0);
  stage01_replace_invader(c, v, pi, episode, 0);
  V3 p0{0, 0, 0};
  for (int s = 0; s < 2; ++s) {
    V3 p = stage01_cube(c, v.env, RNG_SPAWN_PURSUER, (uint32_t)s, episode, 0);
    replace_planes(v, s, p);
    observe_at_rest(v, s, p);
    if (s == 0) p0 = p;
  }
  v.si(TE_D_MUNITION, 0, 0); v.si(TE_D_MUNITION, 1, 0);
  v.esf(TE_E_LAST_DIST, dist(pi, p0));
  v.esi(TE_E_SNAP_MASK, (int)armed_mask(v));
}

template <int FAMILY, class V> TE_DEV void reset_env(const te_config& c, const V& v, bool prepare = true) {
  if (FAMILY == FAM_STAGE01) stage01_reset_env(c, v);
  else if (FAMILY == FAM_STAGE02) stage02_reset_env(c, v);
  else level4_reset_env(c, v, prepare);
}

// ------------------------------------------------------------------------------------------------
// observation (exp03_vFinal_environment.py:200-228)
// ------------------------------------------------------------------------------------------------
struct ObsOut { float* lidar; float* inertial; float* last_action; };
// floats of one own-sphere row: lidar_channels x 13 x 26 (3: distance, flag, time; 2 = the legacy layout without the time plane)
__host__ __device__ inline int lidar_words(const te_config& c) { return (c.lidar_channels == 2 ? 2 : TE_LIDAR_CHANNELS) * TE_LIDAR_CELLS; }
// prev / persist: persistent observation (te_set_persistent_obs) of the register engage kernels: persist != 0 = RECORD the cells patched
// into obs.lidar, prev[i * Npad + env]: i = 0 the count, i = 1.. the cells (u16); the sub-step launch of the next call erases them
struct StepOut { float* reward; uint8_t* done; int32_t* info; ObsOut obs, term; uint16_t* prev; int persist; };

// LidarMath.cartesian_to_spherical + normalize + binning (lidar_math.py:24-34,93-96,128-137)
TE_DEV void lidar_cell(const te_config& c, V3 local, int& cell, float& rhat) {
  float r = norm(local);
  float theta = 0.0f, phi = 0.0f;
  if (r != 0.0f) { theta = acosf(clampf(local.z / r, -1.0f, 1.0f)); phi = atan2f(local.y, local.x); }
  rhat = clampf(r / c.lidar_radius, 0.0f, 1.0f);
  int ti = min(max((int)(theta / kPi * (float)TE_LIDAR_NTHETA), 0), TE_LIDAR_NTHETA - 1);
  int pi = min(max((int)((phi + kPi) / (2.0f * kPi) * (float)TE_LIDAR_NPHI), 0), TE_LIDAR_NPHI - 1);
  cell = ti * TE_LIDAR_NPHI + pi;
}

// Block-parallel precompute (all threads; items are independent so their LDS reads pipeline):
//   agent inverse attitude (LidarMath.reframe: own_q_inverted, lidar_math.py:53-83), pursuer-invader
//   distances, dome / origin flags, and the LIDAR cell + range of every other drone seen from the agent.
// Call between two __syncthreads(); contains one barrier itself.
TE_DEV void precompute_block(const te_config& c, uint32_t* sm, const Rows& r) {
  const int D = r.D, P = r.P, I = r.I();
  const int tid = threadIdx.x, nt = blockDim.x;
  auto pos = [&](int s, int l) {
    return V3{__uint_as_float(sm[(r.obs_pos() + 0 * D + s) * kEPB + l]), __uint_as_float(sm[(r.obs_pos() + 1 * D + s) * kEPB + l]),
              __uint_as_float(sm[(r.obs_pos() + 2 * D + s) * kEPB + l])};
  };
  if (tid < kEPB) {  // wave 0: inverse attitude of the agent, quaternion rebuilt from the IMU euler angles (imu.py:38)
    const int l = tid;
    Q4 q = quat_of_euler(V3{__uint_as_float(sm[(r.agent() + 0) * kEPB + l]), __uint_as_float(sm[(r.agent() + 1) * kEPB + l]),
                            __uint_as_float(sm[(r.agent() + 2) * kEPB + l])});
    float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    M3 R = rotation(Q4{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2});
    const float m[9] = {R.m00, R.m01, R.m02, R.m10, R.m11, R.m12, R.m20, R.m21, R.m22};
#pragma unroll
    for (int k = 0; k < 9; ++k) sm[(r.rinv() + k) * kEPB + l] = __float_as_uint(m[k]);
    uint32_t zone = 0, org = 0, armed = 0;
    for (int s = 0; s < D; ++s) {
      if (!sm[(r.armed() + s) * kEPB + l]) continue;  // every reader masks these bits with the armed snapshot
      armed |= 1u << s;
      float n = norm(pos(s, l));
      zone |= (n > c.dome_radius ? 1u : 0u) << s;
      org |= (n < c.origin_range ? 1u : 0u) << s;
    }
    sm[r.zone() * kEPB + l] = zone; sm[r.origin() * kEPB + l] = org; sm[r.smask() * kEPB + l] = armed;
    sm[r.hitmask() * kEPB + l] = 0u; sm[r.done() * kEPB + l] = 0u;
  } else {           // waves 1..: pursuer-invader distance matrix
    for (int it = tid - kEPB; it < kEPB * P * I; it += nt - kEPB) {
      int l = it & (kEPB - 1), pj = it / kEPB;
      int pp = pj / I, j = P + pj - pp * I;
      if (!sm[(r.armed() + j) * kEPB + l]) continue;  // read only for pairs of the armed snapshot
      sm[(r.dpi() + pj) * kEPB + l] = __float_as_uint(dist(pos(pp, l), pos(j, l)));
    }
  }
  __syncthreads();
  for (int it = tid; it < kEPB * (D - 1); it += nt) {  // LIDAR features (fused_lidar.py:143-217), one per (env, other drone)
    int l = it & (kEPB - 1), j = 1 + it / kEPB;
    if (!sm[(r.armed() + j) * kEPB + l]) continue;  // resolve_hits() looks at armed drones only (and none is armed later)
    M3 R{__uint_as_float(sm[(r.rinv() + 0) * kEPB + l]), __uint_as_float(sm[(r.rinv() + 1) * kEPB + l]), __uint_as_float(sm[(r.rinv() + 2) * kEPB + l]),
         __uint_as_float(sm[(r.rinv() + 3) * kEPB + l]), __uint_as_float(sm[(r.rinv() + 4) * kEPB + l]), __uint_as_float(sm[(r.rinv() + 5) * kEPB + l]),
         __uint_as_float(sm[(r.rinv() + 6) * kEPB + l]), __uint_as_float(sm[(r.rinv() + 7) * kEPB + l]), __uint_as_float(sm[(r.rinv() + 8) * kEPB + l])};
    int cell; float rhat;
    lidar_cell(c, mul(R, sub(pos(j, l), pos(0, l))), cell, rhat);
    sm[(r.lcell() + j) * kEPB + l] = (uint32_t)cell;
    sm[(r.lrhat() + j) * kEPB + l] = __float_as_uint(rhat);
  }
}

// LidarMath.add_features (lidar_math.py:262-311), closer wins with a strict '<' in slot order, over the
// drones armed NOW: bit j of the result = drone j owns its cell.
TE_DEV uint32_t resolve_hits(const SView& v, uint32_t armed_now) {
  uint32_t owners = 0;
  for (uint32_t todo = armed_now & ~1u; todo; todo &= todo - 1) {
    const int j = __ffs(todo) - 1;
    const int cell = (int)v.sm[v.at(v.r.lcell() + j)];
    const float rhat = __uint_as_float(v.sm[v.at(v.r.lrhat() + j)]);
    bool placed = false;
    for (uint32_t m = owners; m; m &= m - 1) {
      int k = __ffs(m) - 1;
      if ((int)v.sm[v.at(v.r.lcell() + k)] != cell) continue;
      if (rhat < __uint_as_float(v.sm[v.at(v.r.lrhat() + k)])) owners = (owners & ~(1u << k)) | (1u << j);
      placed = true;
      break;
    }
    if (!placed && rhat < 1.0f) owners |= 1u << j;  // an empty cell holds 1.0
  }
  return owners;
}
// normalize_inertial_data (level4/components/utils/normalization.py:6-30,61-110) + gun state
template <class V> TE_DEV void inertial_obs(const te_config& c, const V& v, int step, float out[TE_OBS_INERTIAL_WORDS], int s = 0) {
  const float two_pi = 2.0f * kPi;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    out[0 + k] = clampf(v.gf(TE_D_OBS_POS + k, s) / c.dome_radius, -1.0f, 1.0f);
    out[3 + k] = clampf(v.gf(TE_D_OBS_VEL + k, s) / c.max_speed, -1.0f, 1.0f);
    out[6 + k] = clampf(v.gf(TE_D_OBS_EULER + k, s) / kPi, -1.0f, 1.0f);
    out[9 + k] = clampf(v.gf(TE_D_OBS_RATE + k, s) / two_pi, -1.0f, 1.0f);
  }
  float g[3];
  gun_state(c, v.gi(TE_D_MUNITION, s), v.gi(TE_D_LAST_FIRED, s), step, max_munition_of(c, s), g);
  out[12] = g[0]; out[13] = g[1]; out[14] = g[2];
}
TE_DEV void write_obs_rows(const ObsOut& o, int env, const float inertial[TE_OBS_INERTIAL_WORDS], const float la[4]) {
  if (o.inertial) {
#pragma unroll
    for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) o.inertial[(size_t)env * TE_OBS_INERTIAL_WORDS + k] = inertial[k];
  }
  if (o.last_action) reinterpret_cast<float4*>(o.last_action)[env] = make_float4(la[0], la[1], la[2], la[3]);
}

// The [nvalid,3,13,26] tile of a block is all ones except <= D-1 cells per env: stream the ones at memset
// speed (16-byte coalesced stores, no per-element work), barrier, then patch the few hit cells.
TE_DEV void stream_ones(const te_config& c, float* __restrict__ lidar, int env0, int nvalid) {
  float* base = lidar + (size_t)env0 * lidar_words(c);  // 64 * 1014 (or 676) * 4 B per block: 16-byte aligned
  const int total = nvalid * lidar_words(c);
  const int quads = total >> 2;
  const float4 ones = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
  for (int qi = threadIdx.x; qi < quads; qi += blockDim.x) reinterpret_cast<float4*>(base)[qi] = ones;
  for (int f = (quads << 2) + threadIdx.x; f < total; f += blockDim.x) base[f] = 1.0f;
}
// which buffer gets env l's sphere: the main one, or the terminal one when the env auto-reset (0 = none)
TE_DEV void patch_hits(const te_config& c, const uint32_t* sm, const Rows& r, float* __restrict__ lidar,
                       float* __restrict__ t_lidar, int env0, int nvalid) {
  const int D = r.D;
  for (int it = threadIdx.x; it < kEPB * (D - 1); it += blockDim.x) {
    int l = it & (kEPB - 1), j = 1 + it / kEPB;
    if (l >= nvalid || !((sm[r.hitmask() * kEPB + l] >> j) & 1u)) continue;
    float* dst = sm[r.done() * kEPB + l] ? t_lidar : lidar;
    if (!dst) continue;
    dst += (size_t)(env0 + l) * lidar_words(c) + sm[(r.lcell() + j) * kEPB + l];
    dst[0] = __uint_as_float(sm[(r.lrhat() + j) * kEPB + l]);
    dst[TE_LIDAR_CELLS] = (float)(j < r.P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;  // lidar_math.py:305
    if (c.lidar_channels != 2) dst[2 * TE_LIDAR_CELLS] = 0.1f;  // Delta = 1 of a 10-deep ring (perception_snapshot.py:36-37)
  }
}
// terminal tiles of auto-reset envs (rare): ones, to be patched by patch_hits after the barrier
TE_DEV void stream_terminal_ones(const te_config& c, const uint32_t* sm, const Rows& r, float* __restrict__ t_lidar, int env0, int nvalid) {
  for (int l = 0; l < nvalid; ++l) {
    if (!sm[r.done() * kEPB + l]) continue;
    float* base = t_lidar + (size_t)(env0 + l) * lidar_words(c);
    for (int e = threadIdx.x; e < lidar_words(c); e += blockDim.x) base[e] = 1.0f;
  }
}

// Common tail of every family's step: observation of THIS step (to the terminal buffers when the env
// auto-resets), then `finish` (on_step_end / respawn), then the reset + reset observation.
template <int FAMILY, class FinishFn>
TE_DEV void emit_and_finish(const te_config& c, const SView& v, int step, bool term, uint32_t armed_now, const StepOut& o, FinishFn finish) {
  const bool to_terminal = term && c.auto_reset;
  v.sm[v.at(v.r.done())] = to_terminal ? 1u : 0u;
  v.sm[v.at(v.r.anow())] = armed_now; v.sm[v.at(v.r.sstep())] = (uint32_t)step; v.sm[v.at(v.r.sepis())] = (uint32_t)v.egi(TE_E_EPISODE);
  v.sm[v.at(v.r.hitmask())] = resolve_hits(v, armed_now);  // compute_observation happens before on_step_end
  TE_LSTAMP(12);
  if (to_terminal) {  // SB3 VecEnv auto-reset: the terminal observation goes aside (rare: straight from this lane)
    float in[TE_OBS_INERTIAL_WORDS], la[4];
    inertial_obs(c, v, step, in);
#pragma unroll
    for (int k = 0; k < 4; ++k) la[k] = v.egf(TE_E_LAST_ACTION + k);
    write_obs_rows(o.term, v.env, in, la);
  }
  finish();
  TE_LSTAMP(13);
  if (to_terminal) {
    if (FAMILY == FAM_LEVEL4) {  // env record now; the D slots are respawned by the whole block (spawn phase)
      level4_reset_record(c, v);
      v.sm[v.at(v.r.task())] = 1u | (1u << 8);
      v.pre_valid = false;
    } else reset_env<FAMILY>(c, v, false);
  }
  TE_LSTAMP(14);
  // the main inertial / last_action rows (post-reset values for an auto-reset env) and the scripted
  // commands of the next step are produced by the whole block afterwards (emit_rows / prepare phase)
  v.sm[v.at(v.r.prevalid())] = v.pre_valid ? 1u : 0u;
}

// Block-parallel epilogue 1: inertial [nvalid,15] and last_action [nvalid,4] rows from the LDS copy of the
// agent's IMU read / gun / env record, one output word per thread iteration, coalesced.
TE_DEV void emit_rows(const te_config& c, const uint32_t* sm, const Rows& r, const ObsOut& o, int env0, int nvalid,
                      int tid, int nthreads) {
  const int D = r.D;
  if (o.inertial)
    for (int j = tid; j < nvalid * TE_OBS_INERTIAL_WORDS; j += nthreads) {
      const int l = j / TE_OBS_INERTIAL_WORDS, k = j - l * TE_OBS_INERTIAL_WORDS;
      float val;
      if (k < 3) val = clampf(__uint_as_float(sm[(r.obs_pos() + k * D) * kEPB + l]) / c.dome_radius, -1.0f, 1.0f);
      else if (k < 6) val = clampf(__uint_as_float(sm[(r.agent() + 3 + (k - 3)) * kEPB + l]) / c.max_speed, -1.0f, 1.0f);
      else if (k < 9) val = clampf(__uint_as_float(sm[(r.agent() + (k - 6)) * kEPB + l]) / kPi, -1.0f, 1.0f);
      else if (k < 12) val = clampf(__uint_as_float(sm[(r.agent() + 6 + (k - 9)) * kEPB + l]) / (2.0f * kPi), -1.0f, 1.0f);
      else {
        float g[3];
        gun_state(c, (int)sm[r.munition() * kEPB + l], (int)sm[r.last_fired() * kEPB + l],
                  (int)sm[(r.env() + TE_E_STEP) * kEPB + l], max_munition_of(c, 0), g);
        val = g[k - 12];
      }
      o.inertial[(size_t)env0 * TE_OBS_INERTIAL_WORDS + j] = val;
    }
  if (o.last_action)
    for (int j = tid; j < nvalid * 4; j += nthreads)
      o.last_action[(size_t)env0 * 4 + j] = __uint_as_float(sm[(r.env() + TE_E_LAST_ACTION + (j & 3)) * kEPB + (j >> 2)]);
}

// ------------------------------------------------------------------------------------------------
// level4 family: Env.step after advance_step (exp03_vFinal_environment.py:163-171) =
// Task.on_step_middle + compute_info + compute_observation + on_step_end, then SB3 auto-reset.
// ------------------------------------------------------------------------------------------------
TE_DEV void level4_logic(const te_config& c, const SView& v, float4 action, const StepOut& o) {
  const int Pn = c.n_pursuers;
  v.esf(TE_E_LAST_ACTION + 0, action.x); v.esf(TE_E_LAST_ACTION + 1, action.y);
  v.esf(TE_E_LAST_ACTION + 2, action.z); v.esf(TE_E_LAST_ACTION + 3, action.w);
  const int step = v.egi(TE_E_STEP) + 1;  // AGENT_STEP_BROADCAST (exp03_vFinal_environment.py:177-182)
  v.esi(TE_E_STEP, step);
  const uint32_t episode = (uint32_t)v.egi(TE_E_EPISODE);
  // OffsetHandler.on_middle_step snapshot: drones armed now.  S and the dome / origin flags are bit masks from the
  // precompute phase, and A tracks who is STILL armed as the engagement disarms drones, so the per-slot questions
  // below are bit operations instead of LDS round trips in loops the compiler cannot unroll (D is a run-time value).
  const uint32_t S = v.sm[v.at(v.r.smask())];
  const uint32_t zone = v.sm[v.at(v.r.zone())] & S, org = v.sm[v.at(v.r.origin())] & S;
  const uint32_t pur_bits = (1u << Pn) - 1u, inv_bits = ~pur_bits;
  uint32_t A = S;
  auto kill = [&](int j) { disarm(v, j); A &= ~(1u << j); };
  v.esi(TE_E_SNAP_MASK, (int)S);

  // closest invader of every pursuer over the snapshot, once: identify_invaders_in_range(R)[p][0] is that
  // invader whenever its distance is below R (offsets_handler.py:283-309), for both ranges
  int tgt_of[2] = {-1, -1}; float dmin_of[2] = {0.0f, 0.0f};
  int agent_shots = 0, ally_shots = 0, exploded = 0, pursuer_suicided = 0, agent_suicided = 0;
  // process_shoot_range_invaders (exp03_vFinal_task.py:392-413)
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    float dmin;
    int tgt = closest_invader(v, S, p, dmin);
    if (p < 2) { tgt_of[p] = tgt; dmin_of[p] = dmin; }
    if (tgt < 0 || !(dmin < c.shoot_range)) continue;
    int mun = v.gi(TE_D_MUNITION, p), lf = v.gi(TE_D_LAST_FIRED, p);
    if (!(gun_available(c, mun, lf, step) && mun > 0)) continue;  // Gun.can_fire (gun.py:77-81)
    v.si(TE_D_MUNITION, p, mun - 1);
    v.si(TE_D_LAST_FIRED, p, step);
    U4 r = env_rng(c, v.env, RNG_HIT, (uint32_t)p, 0, episode, (uint32_t)step);
    if (u01(r.x) < c.hit_prob) {  // gun.py:94; entities_manager.shoot_by_ids (:238-248)
      kill(tgt);
      if (p == 0) agent_shots += 1; else ally_shots += 1;
      if (c.evaluation) v.si(TE_D_KILLS, p, v.gi(TE_D_KILLS, p) + 1);  // lw_kills (evaluation_task.py:498-499)
    }
  }
  TE_LSTAMP(8);
  // process_explosion_range_invaders (:359-390) on the same (stale) distances
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt; float dmin;
    if (p < 2) { tgt = tgt_of[p]; dmin = dmin_of[p]; } else tgt = closest_invader(v, S, p, dmin);
    if (tgt < 0 || !(dmin < c.explosion_range)) continue;
    kill(p);
    kill(tgt);
    int mun = v.gi(TE_D_MUNITION, p);
    if (mun == 0 && p == 0) agent_suicided += 1;
    else if (mun == 0) pursuer_suicided += 1;
    else exploded += 1;
  }
  const int agent_kills = v.egi(TE_E_AGENT_KILLS) + agent_shots;
  const int allies_kills = v.egi(TE_E_ALLIES_KILLS) + ally_shots;
  const int deads = v.egi(TE_E_DEADS) + exploded;
  v.esi(TE_E_AGENT_KILLS, agent_kills); v.esi(TE_E_ALLIES_KILLS, allies_kills); v.esi(TE_E_DEADS, deads);
  // process_invaders_in_origin (:656-659); commented out in Evaluation_Task.on_step_middle (evaluation_task.py:397)
  if (!c.evaluation || (c.evaluation & TE_EVAL_ORIGIN_RULE))
    for (uint32_t m = org & inv_bits; m; m &= m - 1) kill(__ffs(m) - 1);

  TE_LSTAMP(9);
  // compute_reward (:423-515)
  float reward = 0.0f;  // Evaluation_Task.compute_reward returns 0 (evaluation_task.py:508-515)
  const V3 apos = obs_pos(v, 0);
  if (!c.evaluation) {
    float g[3];
    gun_state(c, v.gi(TE_D_MUNITION, 0), v.gi(TE_D_LAST_FIRED, 0), step, max_munition_of(c, 0), g);
    const float dist_origin = norm(apos);
    // identify_closest_ally / identify_closest_invader (offsets_handler.py:167-190,256-281)
    const int ally = closest_ally(v, S, 0);
    const int from = ally < 0 ? 0 : ally;
    int target = -1;
    if ((S >> from) & 1u) { float dm; target = from < 2 ? tgt_of[from] : closest_invader(v, S, from, dm); }
    V3 tp = target >= 0 ? obs_pos(v, target) : V3{0, 0, 0};
    const float cur = dist(apos, tp);
    const bool ready = g[2] == 1.0f || g[0] == 0.0f;
    float score, bonus = 0.0f, penalty = 0.0f;
    const float last = v.egf(TE_E_LAST_DIST);
    if (0.01f < last - cur && ready)
      bonus += c.approach_bonus_gain * norm(V3{v.gf(TE_D_OBS_VEL, 0), v.gf(TE_D_OBS_VEL + 1, 0), v.gf(TE_D_OBS_VEL + 2, 0)});
    v.esf(TE_E_LAST_DIST, cur);
    score = ready ? -cur : cur * (2.0f * g[1] - 1.0f);
    if (agent_shots > 0 || agent_suicided > 0) bonus += (float)(agent_shots + agent_suicided) * 1000.0f;
    if (ally_shots > 0 || pursuer_suicided > 0) bonus += 0.5f * (float)(ally_shots + pursuer_suicided) * 1000.0f;
    else if (exploded > 0) penalty += 1000.0f * (float)exploded;
    if (apos.z < -5.0f) penalty += (-5.0f - apos.z) * 1000.0f;
    if (zone & pur_bits) penalty += 1000.0f;  // any armed pursuer outside the dome
    if (dist_origin > c.born_radius - 2.0f) penalty += dist_origin - c.born_radius - 2.0f;  // literal
    reward = score + bonus - penalty;
  }
  TE_LSTAMP(10);
  // increment_max_step (:150-153)
  int max_step = v.egi(TE_E_MAX_STEP);
  if (agent_shots + ally_shots > 0) { max_step += c.step_increment; v.esi(TE_E_MAX_STEP, max_step); }
  // compute_termination (:517-569)
  const int armed_invaders = __popc(A & inv_bits), armed_pursuers = __popc(A & pur_bits);
  const int round = v.egi(TE_E_ROUND);
  const bool all_rounds_over = armed_invaders == 0 && round >= c.n_rounds;
  bool term;
  if (c.evaluation) {  // evaluation_task.py:519-551: optional time limit, all rounds over, anybody outside, nobody left
    term = (c.max_step > 0 && step > max_step) || all_rounds_over || zone != 0u || armed_pursuers == 0;
  } else {
    term = step > max_step || all_rounds_over;
    if (!term) {
      if (zone || armed_pursuers == 0 || !(A & 1u) || apos.z < -5.99f) term = true;
    }
  }
  TE_LSTAMP(11);
  // info (:571-578)
  o.reward[v.env] = reward;
  o.done[v.env] = term ? 1 : 0;
  reinterpret_cast<int4*>(o.info)[v.env] = make_int4(agent_kills, allies_kills, deads, round);
  v.esi(TE_E_INFO_WAVE, round);
  emit_and_finish<FAM_LEVEL4>(c, v, step, term, A, o, [&]() {
    // on_step_end (:321-333): next wave when this one is cleared and a pursuer is alive
    if (!term && armed_invaders == 0 && armed_pursuers > 0) {
      int next = round + (round < c.n_rounds ? 1 : c.n_rounds);  // advance_round (:155-175)
      v.esi(TE_E_ROUND, next);
      set_snap_mask(v, level4_mask_after_spawn(c, v, next, false));
      v.sm[v.at(v.r.task())] = (uint32_t)next;  // the D slots are respawned by the whole block (spawn phase)
      v.pre_valid = false;
    }
  });
}

// stage02: L3Stage1.on_step_middle etc. (level3/components/stages.py:144-179,241-344)
TE_DEV void stage02_logic(const te_config& c, const SView& v, float4 action, const StepOut& o) {
  const int Pn = c.n_pursuers, D = v.D;
  v.esf(TE_E_LAST_ACTION + 0, action.x); v.esf(TE_E_LAST_ACTION + 1, action.y);
  v.esf(TE_E_LAST_ACTION + 2, action.z); v.esf(TE_E_LAST_ACTION + 3, action.w);
  const int step = v.egi(TE_E_STEP) + 1;
  v.esi(TE_E_STEP, step);
  const uint32_t episode = (uint32_t)v.egi(TE_E_EPISODE);
  const uint32_t S = armed_mask(v);
  v.esi(TE_E_SNAP_MASK, (int)S);
  const float cur = stage02_agent_min_distance(c, v, S);  // before any respawn invalidates the distance rows
  int shots = 0, exploded = 0;
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    float dmin;
    int tgt = closest_invader(v, S, p, dmin);
    if (tgt < 0 || !(dmin < c.shoot_range)) continue;
    int mun = v.gi(TE_D_MUNITION, p), lf = v.gi(TE_D_LAST_FIRED, p);
    // shoot_by_ids with the suicide rule (level3/components/quadcopter_manager.py:155-171)
    if (mun == 0) { disarm(v, tgt); shots += 1; continue; }
    if (!gun_available(c, mun, lf, step)) continue;
    v.si(TE_D_MUNITION, p, mun - 1);
    v.si(TE_D_LAST_FIRED, p, step);
    U4 r = env_rng(c, v.env, RNG_HIT, (uint32_t)p, 0, episode, (uint32_t)step);
    if (u01(r.x) < c.hit_prob) { disarm(v, tgt); shots += 1; }
  }
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    float dmin;
    int tgt = closest_invader(v, S, p, dmin);
    if (tgt < 0 || !(dmin < c.explosion_range)) continue;
    disarm(v, p); disarm(v, tgt); exploded += 1;
  }
  const int kills = v.egi(TE_E_AGENT_KILLS) + shots, deads = v.egi(TE_E_DEADS) + exploded;
  v.esi(TE_E_AGENT_KILLS, kills); v.esi(TE_E_DEADS, deads);
  float g[3];
  gun_state(c, v.gi(TE_D_MUNITION, 0), v.gi(TE_D_LAST_FIRED, 0), step, max_munition_of(c, 0), g);
  const float last = v.egf(TE_E_PREV_SNAP_MIN);
  float score, bonus = 0.0f, penalty = 0.0f;
  if (g[2] == 1.0f) score = -cur;
  else if (g[0] == 0.0f) score = -cur;
  else score = cur * (2.0f * g[1] - 1.0f);
  if (0.01f < last - cur && (g[2] == 1.0f || g[0] == 0.0f))
    bonus += c.approach_bonus_gain * norm(V3{v.gf(TE_D_OBS_VEL, 0), v.gf(TE_D_OBS_VEL + 1, 0), v.gf(TE_D_OBS_VEL + 2, 0)});
  bonus += 1000.0f * (float)shots;
  penalty += 1000.0f * (float)exploded;
  int outside_p = 0, outside_i = 0;
  for (int s = 0; s < D; ++s)
    if (((S >> s) & 1u) && outside_dome(c, v, s)) { if (s < Pn) outside_p += 1; else outside_i += 1; }
  if (outside_p > 0) penalty += 1000.0f;
  const float reward = score + bonus - penalty;
  int armed_pursuers = 0;
  for (int p = 0; p < Pn; ++p) armed_pursuers += v.gi(TE_D_ARMED, p) ? 1 : 0;
  const bool term = step > v.egi(TE_E_MAX_STEP) || outside_p > 0 || outside_i > 0 || armed_pursuers < Pn;
  o.reward[v.env] = reward;
  o.done[v.env] = term ? 1 : 0;
  reinterpret_cast<int4*>(o.info)[v.env] = make_int4(kills, 0, deads, 0);
  // the observation is taken before the respawn: a drone armed after the step broadcast has no
  // Delta=1 snapshot yet (lidar_buffer.py:443-447) and is invisible this step
  emit_and_finish<FAM_STAGE02>(c, v, step, term, armed_mask(v), o, [&]() {
    for (int j = Pn; j < D; ++j)  // respawn killed invaders (stages.py:167-174)
      if (!v.gi(TE_D_ARMED, j)) respawn_armed(c, v, j, stage02_invader_position(c, v.env, j, episode, (uint32_t)step));
    v.esf(TE_E_PREV_SNAP_MIN, cur); v.esf(TE_E_LAST_DIST, cur);  // on_step_end: last_offsets = current_offsets
  });
}

// stage01: PyflytL2EnviromentModifiedV2.step after the sim loop (pyflyt_level2_environment_modified_v2.py:137-145)
TE_DEV void stage01_logic(const te_config& c, const SView& v, float4 action, const StepOut& o) {
  v.esf(TE_E_LAST_ACTION + 0, action.x); v.esf(TE_E_LAST_ACTION + 1, action.y);
  v.esf(TE_E_LAST_ACTION + 2, action.z); v.esf(TE_E_LAST_ACTION + 3, action.w);
  const int step = v.egi(TE_E_STEP) + 1;  // step_calls
  v.esi(TE_E_STEP, step);
  const uint32_t episode = (uint32_t)v.egi(TE_E_EPISODE);
  const V3 pp = obs_pos(v, 0), pi = obs_pos(v, 2);
  const float d = dist(pi, pp);
  float bonus = 0.0f, penalty = 0.0f;
  if (d < v.egf(TE_E_LAST_DIST))
    bonus += c.approach_bonus_gain * norm(V3{v.gf(TE_D_OBS_VEL, 0), v.gf(TE_D_OBS_VEL + 1, 0), v.gf(TE_D_OBS_VEL + 2, 0)});
  if (d < c.catch_distance) bonus += 1000.0f;
  if (d > c.dome_radius) penalty += 1000.0f;
  const float reward = -d + bonus - penalty;
  const bool term = step > v.egi(TE_E_MAX_STEP) || norm(pp) > c.dome_radius || norm(pi) > c.dome_radius;
  int kills = v.egi(TE_E_AGENT_KILLS) + (d < c.catch_distance ? 1 : 0);
  o.reward[v.env] = reward;
  o.done[v.env] = term ? 1 : 0;
  reinterpret_cast<int4*>(o.info)[v.env] = make_int4(kills, 0, 0, 0);
  emit_and_finish<FAM_STAGE01>(c, v, step, term, armed_mask(v), o, [&]() {
    if (d < c.catch_distance) {  // replace_invader_if_close (:147-154)
      stage01_replace_invader(c, v, stage01_cube(c, v.env, RNG_RESPAWN, 2, episode, (uint32_t)step), episode, (uint32_t)step);
      v.esi(TE_E_AGENT_KILLS, kills);
    }
    v.esf(TE_E_LAST_DIST, dist(obs_pos(v, 2), obs_pos(v, 0)));  // update_last_distance (:219-223)
  });
}

}  // namespace te
