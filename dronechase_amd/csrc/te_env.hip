// te_env.hip — gfx950 kernels + C ABI of the batched threat-engagement environment.
//
// One te_step = two launches on the caller's stream:
//   K1 substeps_kernel       one wavefront per (drone slot, 64 consecutive envs); armed lanes fly the 16
//                            physics sub-steps in registers (state read once, written once); waves whose
//                            slot is disarmed in all 64 envs retire immediately.
//   K2 engage_observe_kernel one 256-thread block per 64 envs: wave 0 resolves engagement, reward,
//                            termination, wave progression, auto-reset and the scripted commands of the
//                            NEXT step (one lane per env); LIDAR hits are binned in LDS and all four
//                            waves then stream the [64,3,13,26] observation tile with 16-byte stores.
// No MFMA: this is element-wise physics and byte streaming.
//
// Reference citations are file:line under the reference's src/ tree.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "te_device.hpp"

namespace te {

constexpr int kEnvsPerBlock = 64;   // K2: envs per block (= one wavefront of logic lanes)
constexpr int kMaxD = 32;
constexpr int kMapStride = 340;     // bytes per env in the LDS cell map (338 cells, padded)
enum Family { FAM_LEVEL4 = 0, FAM_STAGE01 = 1, FAM_STAGE02 = 2 };

// ============================================================================================
// K1: sub-steps
// ============================================================================================
template <int FAMILY, bool NOISE>
__global__ __launch_bounds__(256) void substeps_kernel(Params p, const float* __restrict__ actions) {
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  const int lane = threadIdx.x & 63;
  const int D = p.D;
  const int chunk = wave / D;
  const int slot = wave - chunk * D;
  if (chunk >= (p.Npad >> 6)) return;
  const int env = chunk * 64 + lane;
  if (env >= p.N) return;
  Planes P{p.dstate, p.estate, D, p.Npad, env};
  if (!P.di(TE_D_ARMED, slot)) return;
  const te_config& c = p.cfg;

  // ---- set-point for this env.step
  float sp[4];
  if (slot == 0) {  // RL agent: Quadcopter.drive (quadcopter.py:398-413)
    const float4 a = reinterpret_cast<const float4*>(actions)[env];
    command_to_velocity(a.x, a.y, a.z, a.w, sp[0], sp[1], sp[3]);
    sp[2] = 0.0f;
  } else if (FAMILY == FAM_LEVEL4) {  // scripted drones: command prepared by the previous K2 / reset
    sp[0] = P.df(TE_X_CMD + 0, slot); sp[1] = P.df(TE_X_CMD + 1, slot); sp[2] = 0.0f; sp[3] = P.df(TE_X_CMD + 2, slot);
    if (slot >= c.n_pursuers) P.di(TE_D_NAV_STATE, slot) = P.di(TE_X_NAV_NEXT, slot);
  } else {  // stage01 / stage02: persistent set-points
#pragma unroll
    for (int k = 0; k < 4; ++k) sp[k] = P.df(TE_D_SETPOINT + k, slot);
  }
  if (slot == 0 || FAMILY == FAM_LEVEL4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) P.df(TE_D_SETPOINT + k, slot) = sp[k];
  }

  // ---- load
  Body b;
  b.pos = V3{P.df(TE_D_POS, slot), P.df(TE_D_POS + 1, slot), P.df(TE_D_POS + 2, slot)};
  b.q = Q4{P.df(TE_D_QUAT, slot), P.df(TE_D_QUAT + 1, slot), P.df(TE_D_QUAT + 2, slot), P.df(TE_D_QUAT + 3, slot)};
  b.vel = V3{P.df(TE_D_VEL, slot), P.df(TE_D_VEL + 1, slot), P.df(TE_D_VEL + 2, slot)};
  {
    V3 ww{P.df(TE_D_OMEGA, slot), P.df(TE_D_OMEGA + 1, slot), P.df(TE_D_OMEGA + 2, slot)};
    b.wb = mulT(rotation(b.q), ww);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) b.thr[k] = P.df(TE_D_THROTTLE + k, slot);
#pragma unroll
  for (int k = 0; k < 3; ++k) { b.av_i[k] = P.df(TE_D_PID_AV_I + k, slot); b.av_e[k] = P.df(TE_D_PID_AV_E + k, slot); }
#pragma unroll
  for (int k = 0; k < 2; ++k) { b.lv_i[k] = P.df(TE_D_PID_LV_I + k, slot); b.lv_e[k] = P.df(TE_D_PID_LV_E + k, slot); }
  b.zv_i = P.df(TE_D_PID_ZV_I, slot); b.zv_e = P.df(TE_D_PID_ZV_E, slot);
  V3 pf{0, 0, 0}, pt{0, 0, 0};
  const bool mode7 = (FAMILY == FAM_STAGE01) && slot == 2;
  if (mode7) {
    pf = V3{P.df(TE_D_PENDING, slot), P.df(TE_D_PENDING + 1, slot), P.df(TE_D_PENDING + 2, slot)};
    pt = V3{P.df(TE_D_PENDING + 3, slot), P.df(TE_D_PENDING + 4, slot), P.df(TE_D_PENDING + 5, slot)};
  }
  const uint32_t episode = (uint32_t)P.ei(TE_E_EPISODE);
  // stage01 counts step_calls BEFORE the sim loop (pyflyt_level2_environment_modified_v2.py:128)
  const uint32_t step_index = (uint32_t)P.ei(TE_E_STEP) + (FAMILY == FAM_STAGE01 ? 1u : 0u);

  // ---- fly
  const int S = c.substeps;
  const int last = c.observe_lag ? S - 1 : -1;
  float nz[4] = {0, 0, 0, 0};
  for (int s = 0; s < S; ++s) {
    if (NOISE) motor_noise(c, env, slot, episode, step_index, s, nz);
    if (mode7) substep<true>(c, b, sp, nz, pf, pt, s == last);  // wave-uniform: slot is per wave
    else substep<false>(c, b, sp, nz, pf, pt, s == last);
  }

  // ---- store
  const M3 R = rotation(b.q);
  if (!c.observe_lag) {  // IMU refreshed after the loop
    b.o_pos = b.pos; b.o_vel = mulT(R, b.vel); b.o_rate = b.wb; b.o_eul = euler_of(b.q);
  }
  const V3 ww = mul(R, b.wb);
  P.df(TE_D_POS, slot) = b.pos.x; P.df(TE_D_POS + 1, slot) = b.pos.y; P.df(TE_D_POS + 2, slot) = b.pos.z;
  P.df(TE_D_QUAT, slot) = b.q.x; P.df(TE_D_QUAT + 1, slot) = b.q.y; P.df(TE_D_QUAT + 2, slot) = b.q.z; P.df(TE_D_QUAT + 3, slot) = b.q.w;
  P.df(TE_D_VEL, slot) = b.vel.x; P.df(TE_D_VEL + 1, slot) = b.vel.y; P.df(TE_D_VEL + 2, slot) = b.vel.z;
  P.df(TE_D_OMEGA, slot) = ww.x; P.df(TE_D_OMEGA + 1, slot) = ww.y; P.df(TE_D_OMEGA + 2, slot) = ww.z;
#pragma unroll
  for (int k = 0; k < 4; ++k) P.df(TE_D_THROTTLE + k, slot) = b.thr[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { P.df(TE_D_PID_AV_I + k, slot) = b.av_i[k]; P.df(TE_D_PID_AV_E + k, slot) = b.av_e[k]; }
#pragma unroll
  for (int k = 0; k < 2; ++k) { P.df(TE_D_PID_LV_I + k, slot) = b.lv_i[k]; P.df(TE_D_PID_LV_E + k, slot) = b.lv_e[k]; }
  P.df(TE_D_PID_ZV_I, slot) = b.zv_i; P.df(TE_D_PID_ZV_E, slot) = b.zv_e;
  P.df(TE_D_OBS_POS, slot) = b.o_pos.x; P.df(TE_D_OBS_POS + 1, slot) = b.o_pos.y; P.df(TE_D_OBS_POS + 2, slot) = b.o_pos.z;
  P.df(TE_D_OBS_EULER, slot) = b.o_eul.x; P.df(TE_D_OBS_EULER + 1, slot) = b.o_eul.y; P.df(TE_D_OBS_EULER + 2, slot) = b.o_eul.z;
  P.df(TE_D_OBS_VEL, slot) = b.o_vel.x; P.df(TE_D_OBS_VEL + 1, slot) = b.o_vel.y; P.df(TE_D_OBS_VEL + 2, slot) = b.o_vel.z;
  P.df(TE_D_OBS_RATE, slot) = b.o_rate.x; P.df(TE_D_OBS_RATE + 1, slot) = b.o_rate.y; P.df(TE_D_OBS_RATE + 2, slot) = b.o_rate.z;
  if (mode7) {
#pragma unroll
    for (int k = 0; k < 6; ++k) P.df(TE_D_PENDING + k, slot) = 0.0f;
  }
}

// ============================================================================================
// entity operations on planes (quadcopter.py:433-478, gun.py)
// ============================================================================================
TE_DEV V3 obs_pos(const Planes& P, int s) { return V3{P.df(TE_D_OBS_POS, s), P.df(TE_D_OBS_POS + 1, s), P.df(TE_D_OBS_POS + 2, s)}; }
TE_DEV float dist(V3 a, V3 b) { return norm(sub(a, b)); }

// Quadcopter.disarm (quadcopter.py:461-478): static body, velocities zeroed, motors/body/set-point/pwm
// reset.  PID memories and the last IMU read stay.
TE_DEV void disarm(const Planes& P, int s) {
  P.di(TE_D_ARMED, s) = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { P.df(TE_D_VEL + k, s) = 0.0f; P.df(TE_D_OMEGA + k, s) = 0.0f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { P.df(TE_D_THROTTLE + k, s) = 0.0f; P.df(TE_D_SETPOINT + k, s) = 0.0f; }
}
// IMU read from the stored world state (imu.py:27-41)
TE_DEV void observe_planes(const Planes& P, int s) {
  Q4 q{P.df(TE_D_QUAT, s), P.df(TE_D_QUAT + 1, s), P.df(TE_D_QUAT + 2, s), P.df(TE_D_QUAT + 3, s)};
  M3 R = rotation(q);
  V3 vb = mulT(R, V3{P.df(TE_D_VEL, s), P.df(TE_D_VEL + 1, s), P.df(TE_D_VEL + 2, s)});
  V3 wb = mulT(R, V3{P.df(TE_D_OMEGA, s), P.df(TE_D_OMEGA + 1, s), P.df(TE_D_OMEGA + 2, s)});
  V3 e = euler_of(q);
  P.df(TE_D_OBS_POS, s) = P.df(TE_D_POS, s); P.df(TE_D_OBS_POS + 1, s) = P.df(TE_D_POS + 1, s); P.df(TE_D_OBS_POS + 2, s) = P.df(TE_D_POS + 2, s);
  P.df(TE_D_OBS_EULER, s) = e.x; P.df(TE_D_OBS_EULER + 1, s) = e.y; P.df(TE_D_OBS_EULER + 2, s) = e.z;
  P.df(TE_D_OBS_VEL, s) = vb.x; P.df(TE_D_OBS_VEL + 1, s) = vb.y; P.df(TE_D_OBS_VEL + 2, s) = vb.z;
  P.df(TE_D_OBS_RATE, s) = wb.x; P.df(TE_D_OBS_RATE + 1, s) = wb.y; P.df(TE_D_OBS_RATE + 2, s) = wb.z;
}
// Quadcopter.replace (quadcopter.py:433-439): teleport, identity attitude, zero base velocity
TE_DEV void replace_planes(const Planes& P, int s, V3 p) {
  P.df(TE_D_POS, s) = p.x; P.df(TE_D_POS + 1, s) = p.y; P.df(TE_D_POS + 2, s) = p.z;
  P.df(TE_D_FORMATION, s) = p.x; P.df(TE_D_FORMATION + 1, s) = p.y; P.df(TE_D_FORMATION + 2, s) = p.z;
  P.df(TE_D_QUAT, s) = 0.0f; P.df(TE_D_QUAT + 1, s) = 0.0f; P.df(TE_D_QUAT + 2, s) = 0.0f; P.df(TE_D_QUAT + 3, s) = 1.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { P.df(TE_D_VEL + k, s) = 0.0f; P.df(TE_D_OMEGA + k, s) = 0.0f; }
}
// IMU of a drone that has just been teleported (identity attitude, at rest)
TE_DEV void observe_at_rest(const Planes& P, int s, V3 p) {
  P.df(TE_D_OBS_POS, s) = p.x; P.df(TE_D_OBS_POS + 1, s) = p.y; P.df(TE_D_OBS_POS + 2, s) = p.z;
#pragma unroll
  for (int k = 0; k < 3; ++k) { P.df(TE_D_OBS_EULER + k, s) = 0.0f; P.df(TE_D_OBS_VEL + k, s) = 0.0f; P.df(TE_D_OBS_RATE + k, s) = 0.0f; }
}
TE_DEV int max_munition_of(const te_config& c, int slot) {
  if (slot >= c.n_pursuers) return 10;  // Gun default (gun.py:11)
  if (c.task == TE_TASK_STAGE02) return slot == 0 ? c.munition : 10;  // stages.py:118
  return c.munition;
}
// disarm -> replace -> arm of a drone that ends up armed at p (Task.setup_round,
// exp03_vFinal_task.py:180-196): arm() reads the IMU and resets the gun (quadcopter.py:445-459)
TE_DEV void respawn_armed(const te_config& c, const Planes& P, int s, V3 p) {
  disarm(P, s);
  replace_planes(P, s, p);
  observe_at_rest(P, s, p);
  P.di(TE_D_ARMED, s) = 1;
  P.di(TE_D_MUNITION, s) = max_munition_of(c, s);
  P.di(TE_D_LAST_FIRED, s) = -c.cooldown_steps;
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
TE_DEV uint32_t armed_mask(const Planes& P) {
  uint32_t m = 0;
  for (int s = 0; s < P.D; ++s) m |= (P.di(TE_D_ARMED, s) ? 1u : 0u) << s;
  return m;
}

// spawn samplers -----------------------------------------------------------------------------
// Task.generate_positions (exp03_vFinal_task.py:584-608)
TE_DEV V3 level4_position(const te_config& c, float r, float u_theta, float u_phi) {
  float theta = u_theta * kPi;
  float lower = fminf(c.born_min_z, r);
  float min_phi = acosf(lower / r);
  float phi = (r >= c.born_min_z) ? min_phi + u_phi * (0.5f * kPi - min_phi) : u_phi * (0.5f * kPi);
  float sph, cph, sth, cth;
  sincosf(phi, &sph, &cph); sincosf(theta, &sth, &cth);
  return V3{r * sph * cth, r * sph * sth, r * cph};
}
// L3Stage1.generate_positions (level3/components/stages.py:360-376)
TE_DEV V3 stage02_position(float r, float r_max, float u_r, float u_theta, float u_phi) {
  if (r > r_max) r_max = r;
  float radius = r + u_r * (r_max - r);
  float theta = u_theta * 2.0f * kPi, phi = u_phi * kPi * 0.5f;
  float sph, cph, sth, cth;
  sincosf(phi, &sph, &cph); sincosf(theta, &sth, &cth);
  return V3{radius * sph * cth, radius * sph * sth, radius * cph};
}

// ============================================================================================
// offsets over the snapshot mask (level4/components/entities_management/offsets_handler.py)
// ============================================================================================
TE_DEV int closest_in(const Planes& P, uint32_t mask, int lo, int hi, V3 from, int skip) {
  int best = -1; float bd = 0.0f;
  for (int s = lo; s < hi; ++s) {
    if (s == skip || !((mask >> s) & 1u)) continue;
    float d = dist(obs_pos(P, s), from);
    if (best < 0 || d < bd) { best = s; bd = d; }
  }
  return best;
}

// ============================================================================================
// scripted commands of the NEXT step (Task.on_step_start, exp03_vFinal_task.py:232-244,276-283).
// They depend only on the state at the end of this step, so they are prepared here and consumed by K1.
// ============================================================================================
TE_DEV void set_cmd_toward(const Planes& P, int s, V3 from, V3 to, float speed) {
  V3 v = sub(to, from);
  float n = norm(v);
  float inv = n > 0.0f ? 1.0f / n : 1.0f;  // zero vector stays zero (…air_combat_only.py:191-195)
  float vx, vy, vz;
  command_to_velocity(v.x * inv, v.y * inv, v.z * inv, speed, vx, vy, vz);
  P.df(TE_X_CMD + 0, s) = vx; P.df(TE_X_CMD + 1, s) = vy; P.df(TE_X_CMD + 2, s) = vz;
}
// GeometryUtils.is_point_inside_cone (geometry_utils.py:6-29)
TE_DEV bool inside_cone(V3 p, V3 apex, V3 base, float degrees) {
  V3 ab = sub(base, apex), ap = sub(p, apex);
  float nab = norm(ab), nap = norm(ap);
  if (nap > nab) return false;
  float cosang = (ap.x * ab.x + ap.y * ab.y + ap.z * ab.z) / (nap * nab);
  return acosf(cosang) * (180.0f / kPi) <= 0.5f * degrees;
}
TE_DEV bool building_path_clear(const te_config& c, const Planes& P, uint32_t mask, int s, float degrees) {
  if (!c.kamikaze_cone_check) return false;  // …air_combat_only.py:83-96: constant False
  V3 b{c.building_position[0], c.building_position[1], c.building_position[2]};
  V3 me = obs_pos(P, s);
  for (int p = 0; p < c.n_pursuers; ++p)
    if (((mask >> p) & 1u) && inside_cone(obs_pos(P, p), me, b, degrees)) return false;
  return true;
}
TE_DEV void prepare_level4_commands(const te_config& c, const Planes& P) {
  const int Pn = c.n_pursuers, D = P.D;
  const uint32_t S = (uint32_t)P.ei(TE_E_SNAP_MASK);
  const int step = P.ei(TE_E_STEP);
  const uint32_t pursuer_bits = S & ((1u << Pn) - 1u);
  // KamikazeNavigator.update (…air_combat_only.py:68-78): transition registered, OLD state executes
  for (int j = Pn; j < D; ++j) {
    if (!P.di(TE_D_ARMED, j)) continue;
    int state = P.di(TE_D_NAV_STATE, j), next = state;
    V3 me = obs_pos(P, j);
    if (state == TE_NAV_WAIT) {
      if (building_path_clear(c, P, S, j, 60.0f)) next = TE_NAV_COLLIDE_BUILDING;
      else if (pursuer_bits) next = TE_NAV_COLLIDE_WINGMAN;
      P.df(TE_X_CMD + 0, j) = 0.0f; P.df(TE_X_CMD + 1, j) = 0.0f; P.df(TE_X_CMD + 2, j) = 0.0f;  // hover (:163)
    } else if (state == TE_NAV_COLLIDE_WINGMAN) {
      if (!pursuer_bits) next = TE_NAV_COLLIDE_BUILDING;
      int t = closest_in(P, S, 0, Pn, me, -1);
      V3 target = t >= 0 ? obs_pos(P, t) : V3{0, 0, 0};
      set_cmd_toward(P, j, me, target, c.invader_speed);
    } else {
      if (!building_path_clear(c, P, S, j, 45.0f)) next = TE_NAV_COLLIDE_WINGMAN;
      set_cmd_toward(P, j, me, V3{c.building_position[0], c.building_position[1], c.building_position[2]}, c.invader_speed);
    }
    P.di(TE_X_NAV_NEXT, j) = next;
  }
  // drive_loyalwingmen: get_armed_pursuers()[1:] (exp03_vFinal_task.py:238-244)
  const bool agent_armed = P.di(TE_D_ARMED, 0) != 0;
  for (int a = agent_armed ? 1 : 2; a < Pn; ++a) {
    if (!P.di(TE_D_ARMED, a)) continue;
    if (c.ally_policy == TE_ALLY_BT) {  // LoyalWingmanBehaviorTree (loyalwingman_navigator.py:238-352)
      V3 me = obs_pos(P, a);
      if (gun_available(c, P.di(TE_D_MUNITION, a), P.di(TE_D_LAST_FIRED, a), step)) {
        int t = ((S >> a) & 1u) ? closest_in(P, S, Pn, D, me, -1) : -1;
        V3 target = t >= 0 ? obs_pos(P, t) : V3{0, 0, 0};
        set_cmd_toward(P, a, me, target, c.ally_speed);
      } else {
        set_cmd_toward(P, a, me, V3{P.df(TE_D_FORMATION, a), P.df(TE_D_FORMATION + 1, a), P.df(TE_D_FORMATION + 2, a)}, c.ally_speed);
      }
    } else if (c.ally_policy == TE_ALLY_FROZEN) {  // exp04_vFinal_task.py:240-242: drive([0,0,0,1])
      P.df(TE_X_CMD + 0, a) = 0.0f; P.df(TE_X_CMD + 1, a) = 0.0f; P.df(TE_X_CMD + 2, a) = 0.0f;
    } else {
      P.df(TE_X_CMD + 0, a) = P.df(TE_D_SETPOINT + 0, a); P.df(TE_X_CMD + 1, a) = P.df(TE_D_SETPOINT + 1, a);
      P.df(TE_X_CMD + 2, a) = P.df(TE_D_SETPOINT + 3, a);
    }
  }
}

// ============================================================================================
// resets
// ============================================================================================
// Task.setup_round (exp03_vFinal_task.py:180-196)
TE_DEV void level4_setup_round(const te_config& c, const Planes& P, int round, uint32_t episode) {
  const int Pn = c.n_pursuers;
  for (int j = Pn; j < P.D; ++j) disarm(P, j);
  for (int i = 0; i < round && i < c.n_invaders; ++i) {
    U4 r = env_rng(c, P.env, RNG_SPAWN_INVADER, (uint32_t)(Pn + i), 0, episode, (uint32_t)round);
    respawn_armed(c, P, Pn + i, level4_position(c, c.born_radius, u01(r.x), u01(r.y)));
  }
}
TE_DEV void level4_refresh_snapshot(const Planes& P) {
  P.ei(TE_E_SNAP_MASK) = (int32_t)armed_mask(P);
  for (int s = 0; s < P.D; ++s) P.di(TE_D_NAV_STATE, s) = TE_NAV_WAIT;  // navigators reset()
}
// Env.reset -> Task.on_reset (exp03_vFinal_environment.py:128-146, exp03_vFinal_task.py:255-274)
TE_DEV void level4_reset_env(const te_config& c, const Planes& P) {
  const uint32_t episode = (uint32_t)(P.ei(TE_E_EPISODE) + 1);
  P.ei(TE_E_EPISODE) = (int32_t)episode;
  P.ei(TE_E_STEP) = 0; P.ei(TE_E_MAX_STEP) = c.max_step; P.ei(TE_E_ROUND) = 1;
  P.ei(TE_E_AGENT_KILLS) = 0; P.ei(TE_E_ALLIES_KILLS) = 0; P.ei(TE_E_DEADS) = 0;
  P.ef(TE_E_LAST_DIST) = c.dome_radius;
#pragma unroll
  for (int k = 0; k < 4; ++k) P.ef(TE_E_LAST_ACTION + k) = 0.0f;
  for (int p = 0; p < c.n_pursuers; ++p) disarm(P, p);
  level4_setup_round(c, P, 1, episode);
  for (int p = 0; p < c.n_pursuers; ++p) {
    U4 r = env_rng(c, P.env, RNG_SPAWN_PURSUER, (uint32_t)p, 0, episode, 0);
    respawn_armed(c, P, p, level4_position(c, c.pursuer_spawn_radius, u01(r.x), u01(r.y)));
  }
  level4_refresh_snapshot(P);
  prepare_level4_commands(c, P);
}

// stage02 ------------------------------------------------------------------------------------
TE_DEV float stage02_agent_min_distance(const te_config& c, const Planes& P, uint32_t S) {
  int first = -1;
  for (int p = 0; p < c.n_pursuers; ++p) if ((S >> p) & 1u) { first = p; break; }
  if (first < 0) return 0.0f;
  V3 me = obs_pos(P, first);
  float best = 0.0f; bool any = false;
  for (int j = c.n_pursuers; j < P.D; ++j) {
    if (!((S >> j) & 1u)) continue;
    float d = dist(me, obs_pos(P, j));
    if (!any || d < best) { best = d; any = true; }
  }
  return best;
}
TE_DEV V3 stage02_invader_position(const te_config& c, const Planes& P, int slot, uint32_t episode, uint32_t tag) {
  U4 r = env_rng(c, P.env, RNG_RESPAWN, (uint32_t)slot, 0, episode, tag);
  return stage02_position(2.0f, 6.0f, u01(r.x), u01(r.y), u01(r.z));  // stages.py:378-384
}
// L3Stage1.on_reset (stages.py:104-131)
TE_DEV void stage02_reset_env(const te_config& c, const Planes& P) {
  const uint32_t episode = (uint32_t)(P.ei(TE_E_EPISODE) + 1);
  P.ei(TE_E_EPISODE) = (int32_t)episode;
  P.ei(TE_E_STEP) = 0; P.ei(TE_E_MAX_STEP) = c.max_step; P.ei(TE_E_ROUND) = 0;
  P.ei(TE_E_AGENT_KILLS) = 0; P.ei(TE_E_ALLIES_KILLS) = 0; P.ei(TE_E_DEADS) = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) P.ef(TE_E_LAST_ACTION + k) = 0.0f;
  for (int j = c.n_pursuers; j < P.D; ++j) respawn_armed(c, P, j, stage02_invader_position(c, P, j, episode, 0u));
  for (int p = 0; p < c.n_pursuers; ++p) {
    U4 r = env_rng(c, P.env, RNG_SPAWN_PURSUER, (uint32_t)p, 0, episode, 0);
    respawn_armed(c, P, p, stage02_position(c.pursuer_spawn_radius, 0.0f, u01(r.x), u01(r.y), u01(r.z)));
  }
  uint32_t S = armed_mask(P);
  P.ei(TE_E_SNAP_MASK) = (int32_t)S;
  float d0 = stage02_agent_min_distance(c, P, S);
  P.ef(TE_E_PREV_SNAP_MIN) = d0; P.ef(TE_E_LAST_DIST) = d0;
}

// stage01 ------------------------------------------------------------------------------------
TE_DEV V3 stage01_cube(const te_config& c, const Planes& P, uint32_t purpose, uint32_t slot, uint32_t episode, uint32_t index) {
  U4 r = env_rng(c, P.env, purpose, slot, 0, episode, index);
  return V3{-1.0f + 2.0f * u01(r.x), -1.0f + 2.0f * u01(r.y), -1.0f + 2.0f * u01(r.z)};
}
// QuadcopterManager.replace_invader (level2/components/quadcopter_manager.py:166-179): teleport, raw mode-7
// set-point, then ONE extra imu/control/physics update whose wrench stays accumulated until the next
// stepSimulation.
TE_DEV void stage01_replace_invader(const te_config& c, const Planes& P, V3 p, uint32_t episode, uint32_t step_index) {
  const int s = 2;
  replace_planes(P, s, p);
  observe_at_rest(P, s, p);
  float sp[4] = {p.x, p.y, 0.0f, p.z};
#pragma unroll
  for (int k = 0; k < 4; ++k) P.df(TE_D_SETPOINT + k, s) = sp[k];
  Body b;
  b.pos = p; b.q = Q4{0, 0, 0, 1}; b.vel = V3{0, 0, 0}; b.wb = V3{0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) b.thr[k] = P.df(TE_D_THROTTLE + k, s);
#pragma unroll
  for (int k = 0; k < 3; ++k) { b.av_i[k] = P.df(TE_D_PID_AV_I + k, s); b.av_e[k] = P.df(TE_D_PID_AV_E + k, s); }
#pragma unroll
  for (int k = 0; k < 2; ++k) { b.lv_i[k] = P.df(TE_D_PID_LV_I + k, s); b.lv_e[k] = P.df(TE_D_PID_LV_E + k, s); }
  b.zv_i = P.df(TE_D_PID_ZV_I, s); b.zv_e = P.df(TE_D_PID_ZV_E, s);
  float nz[4] = {0, 0, 0, 0};
  if (c.motor_noise) motor_noise(c, P.env, s, episode, step_index, 255, nz);
  // run controller + motors only: integrate() must not move the body, so evaluate the wrench by
  // differencing a throw-away sub-step (velocity change * mass / dt = applied force, etc.)
  Body t = b;
  V3 pf{0, 0, 0}, pt{0, 0, 0};
  substep<true>(c, t, sp, nz, pf, pt, false);
  const float dt = c.physics_dt;
  V3 F{t.vel.x * c.quad.mass / dt, t.vel.y * c.quad.mass / dt, (t.vel.z / dt + c.quad.gravity) * c.quad.mass};
  V3 Tq{t.wb.x * c.quad.inertia[0] / dt, t.wb.y * c.quad.inertia[1] / dt, t.wb.z * c.quad.inertia[2] / dt};
  P.df(TE_D_PENDING + 0, s) += F.x; P.df(TE_D_PENDING + 1, s) += F.y; P.df(TE_D_PENDING + 2, s) += F.z;
  P.df(TE_D_PENDING + 3, s) += Tq.x; P.df(TE_D_PENDING + 4, s) += Tq.y; P.df(TE_D_PENDING + 5, s) += Tq.z;
#pragma unroll
  for (int k = 0; k < 4; ++k) P.df(TE_D_THROTTLE + k, s) = t.thr[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { P.df(TE_D_PID_AV_I + k, s) = t.av_i[k]; P.df(TE_D_PID_AV_E + k, s) = t.av_e[k]; }
#pragma unroll
  for (int k = 0; k < 2; ++k) { P.df(TE_D_PID_LV_I + k, s) = t.lv_i[k]; P.df(TE_D_PID_LV_E + k, s) = t.lv_e[k]; }
  P.df(TE_D_PID_ZV_I, s) = t.zv_i; P.df(TE_D_PID_ZV_E, s) = t.zv_e;
}
// PyflytL2EnviromentModifiedV2.reset (pyflyt_level2_environment_modified_v2.py:83-123)
TE_DEV void stage01_reset_env(const te_config& c, const Planes& P) {
  const uint32_t episode = (uint32_t)(P.ei(TE_E_EPISODE) + 1);
  P.ei(TE_E_EPISODE) = (int32_t)episode;
  P.ei(TE_E_STEP) = 0; P.ei(TE_E_MAX_STEP) = c.max_step; P.ei(TE_E_ROUND) = 0;
  P.ei(TE_E_AGENT_KILLS) = 0; P.ei(TE_E_ALLIES_KILLS) = 0; P.ei(TE_E_DEADS) = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) P.ef(TE_E_LAST_ACTION + k) = 0.0f;
  for (int s = 0; s < P.D; ++s)
    if (!P.di(TE_D_ARMED, s)) { P.di(TE_D_ARMED, s) = 1; P.di(TE_D_MUNITION, s) = 0; P.di(TE_D_LAST_FIRED, s) = -c.cooldown_steps; }
  V3 pi = stage01_cube(c, P, RNG_SPAWN_INVADER, 2, episode, 0);
  stage01_replace_invader(c, P, pi, episode, 0);
  V3 p0{0, 0, 0};
  for (int s = 0; s < 2; ++s) {
    V3 p = stage01_cube(c, P, RNG_SPAWN_PURSUER, (uint32_t)s, episode, 0);
    replace_planes(P, s, p);
    observe_at_rest(P, s, p);
    if (s == 0) p0 = p;
  }
  P.di(TE_D_MUNITION, 0) = 0; P.di(TE_D_MUNITION, 1) = 0;
  P.ef(TE_E_LAST_DIST) = dist(pi, p0);
  P.ei(TE_E_SNAP_MASK) = (int32_t)armed_mask(P);
}

template <int FAMILY>
TE_DEV void reset_env(const te_config& c, const Planes& P) {
  if (FAMILY == FAM_STAGE01) stage01_reset_env(c, P);
  else if (FAMILY == FAM_STAGE02) stage02_reset_env(c, P);
  else level4_reset_env(c, P);
}

template <int FAMILY>
__global__ __launch_bounds__(256) void reset_kernel(Params p, const uint8_t* __restrict__ mask) {
  int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= p.N) return;
  if (mask && !mask[env]) return;
  Planes P{p.dstate, p.estate, p.D, p.Npad, env};
  reset_env<FAMILY>(p.cfg, P);
}
// recompute the pending scripted commands from a freshly loaded state blob (te_set_state)
__global__ __launch_bounds__(256) void prepare_commands_kernel(Params p) {
  int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= p.N) return;
  Planes P{p.dstate, p.estate, p.D, p.Npad, env};
  prepare_level4_commands(p.cfg, P);
}

// ============================================================================================
// observation (exp03_vFinal_environment.py:200-228): LIDAR hits into LDS, inertial + last action
// ============================================================================================
struct ObsOut {
  float* lidar; float* inertial; float* last_action;
};
struct TileShared {
  uint8_t cellmap[kEnvsPerBlock * kMapStride];  // hit index per (env, cell), 0xFF = empty
  float hit_r[kEnvsPerBlock * kMaxD];           // r_hat of hit h of env lane l at [h * 64 + l]
  float hit_flag[kEnvsPerBlock * kMaxD];
  uint8_t done[kEnvsPerBlock];
  uint8_t nhits[kEnvsPerBlock];
};

// FusedLIDAR.update_data own sphere of the agent (fused_lidar.py:143-217; lidar_math.py:24-34,53-83,93-96,
// 262-311): every OTHER armed drone at its last IMU read, closer wins, flag = type/5, time = 1/10.
TE_DEV void lidar_hits(const te_config& c, const Planes& P, TileShared& sh, int l) {
  V3 own = obs_pos(P, 0);
  Q4 q = quat_of_euler(V3{P.df(TE_D_OBS_EULER, 0), P.df(TE_D_OBS_EULER + 1, 0), P.df(TE_D_OBS_EULER + 2, 0)});
  float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  M3 Rinv = rotation(Q4{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2});
  int nh = 0;
  for (int j = 1; j < P.D; ++j) {
    if (!P.di(TE_D_ARMED, j)) continue;
    V3 local = mul(Rinv, sub(obs_pos(P, j), own));
    float r = norm(local);
    float theta = 0.0f, phi = 0.0f;
    if (r != 0.0f) { theta = acosf(clampf(local.z / r, -1.0f, 1.0f)); phi = atan2f(local.y, local.x); }
    float rhat = clampf(r / c.lidar_radius, 0.0f, 1.0f);
    int ti = min(max((int)(theta / kPi * (float)TE_LIDAR_NTHETA), 0), TE_LIDAR_NTHETA - 1);
    int pi = min(max((int)((phi + kPi) / (2.0f * kPi) * (float)TE_LIDAR_NPHI), 0), TE_LIDAR_NPHI - 1);
    int cell = ti * TE_LIDAR_NPHI + pi;
    uint8_t m = sh.cellmap[l * kMapStride + cell];
    float cur = m == 0xFF ? 1.0f : sh.hit_r[m * kEnvsPerBlock + l];
    if (rhat < cur) {
      sh.cellmap[l * kMapStride + cell] = (uint8_t)nh;
      sh.hit_r[nh * kEnvsPerBlock + l] = rhat;
      sh.hit_flag[nh * kEnvsPerBlock + l] = (float)(j < c.n_pursuers ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;
      ++nh;
    }
  }
  sh.nhits[l] = (uint8_t)nh;
}
// normalize_inertial_data (level4/components/utils/normalization.py:6-30,61-110) + gun state
TE_DEV void write_inertial(const te_config& c, const Planes& P, int step, float* out /* 15 */) {
  const float two_pi = 2.0f * kPi;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    out[0 + k] = clampf(P.df(TE_D_OBS_POS + k, 0) / c.dome_radius, -1.0f, 1.0f);
    out[3 + k] = clampf(P.df(TE_D_OBS_VEL + k, 0) / c.max_speed, -1.0f, 1.0f);
    out[6 + k] = clampf(P.df(TE_D_OBS_EULER + k, 0) / kPi, -1.0f, 1.0f);
    out[9 + k] = clampf(P.df(TE_D_OBS_RATE + k, 0) / two_pi, -1.0f, 1.0f);
  }
  float g[3];
  gun_state(c, P.di(TE_D_MUNITION, 0), P.di(TE_D_LAST_FIRED, 0), step, max_munition_of(c, 0), g);
  out[12] = g[0]; out[13] = g[1]; out[14] = g[2];
}

TE_DEV float tile_value(const TileShared& sh, int l, int rem) {
  int ch = rem >= 2 * TE_LIDAR_CELLS ? 2 : (rem >= TE_LIDAR_CELLS ? 1 : 0);
  int cell = rem - ch * TE_LIDAR_CELLS;
  uint8_t m = sh.cellmap[l * kMapStride + cell];
  if (m == 0xFF) return 1.0f;
  return ch == 0 ? sh.hit_r[m * kEnvsPerBlock + l] : (ch == 1 ? sh.hit_flag[m * kEnvsPerBlock + l] : 0.1f);
}
// Stream the block's [nvalid, 3, 13, 26] float tile (16-byte aligned base) with coalesced float4 stores.
// `blank_done`: envs flagged done get the empty sphere (their real sphere goes to the terminal buffer).
TE_DEV void write_tiles(const TileShared& sh, float* __restrict__ lidar, int env0, int nvalid, bool blank_done) {
  float* base = lidar + (size_t)env0 * TE_OBS_LIDAR_WORDS;
  const int total = nvalid * TE_OBS_LIDAR_WORDS;
  const int quads = total >> 2;
  for (int qi = threadIdx.x; qi < quads; qi += blockDim.x) {
    int f = qi << 2;
    int l = f / TE_OBS_LIDAR_WORDS;
    int rem = f - l * TE_OBS_LIDAR_WORDS;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int ll = l, rr = rem + k;
      if (rr >= TE_OBS_LIDAR_WORDS) { rr -= TE_OBS_LIDAR_WORDS; ll += 1; }
      v[k] = (sh.nhits[ll] == 0 || (blank_done && sh.done[ll])) ? 1.0f : tile_value(sh, ll, rr);
    }
    reinterpret_cast<float4*>(base)[qi] = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int f = (quads << 2) + threadIdx.x; f < total; f += blockDim.x) {
    int l = f / TE_OBS_LIDAR_WORDS;
    int rem = f - l * TE_OBS_LIDAR_WORDS;
    base[f] = (sh.nhits[l] == 0 || (blank_done && sh.done[l])) ? 1.0f : tile_value(sh, l, rem);
  }
}
TE_DEV void write_terminal_tiles(const TileShared& sh, float* __restrict__ t_lidar, int env0, int nvalid) {
  for (int l = 0; l < nvalid; ++l) {
    if (!sh.done[l]) continue;
    float* base = t_lidar + (size_t)(env0 + l) * TE_OBS_LIDAR_WORDS;
    for (int r = threadIdx.x; r < TE_OBS_LIDAR_WORDS; r += blockDim.x)
      base[r] = sh.nhits[l] == 0 ? 1.0f : tile_value(sh, l, r);
  }
}
TE_DEV void clear_maps(TileShared& sh) {
  uint32_t* w = reinterpret_cast<uint32_t*>(sh.cellmap);
  for (int i = threadIdx.x; i < kEnvsPerBlock * kMapStride / 4; i += blockDim.x) w[i] = 0xFFFFFFFFu;
  if (threadIdx.x < kEnvsPerBlock) { sh.done[threadIdx.x] = 0; sh.nhits[threadIdx.x] = 0; }
}

// ============================================================================================
// K2 logic per family (one lane per env)
// ============================================================================================
struct StepOut {
  float* reward; uint8_t* done; int32_t* info;
  ObsOut obs, term;
};

// closest in-range invader of pursuer p over the snapshot S (identify_invaders_in_range()[p][0],
// offsets_handler.py:283-309)
TE_DEV int closest_invader_in_range(const te_config& c, const Planes& P, uint32_t S, int p, float range) {
  V3 me = obs_pos(P, p);
  int best = -1; float bd = 0.0f;
  for (int j = c.n_pursuers; j < P.D; ++j) {
    if (!((S >> j) & 1u)) continue;
    float d = dist(me, obs_pos(P, j));
    if (d < range && (best < 0 || d < bd)) { best = j; bd = d; }
  }
  return best;
}

TE_DEV void emit_obs(const te_config& c, const Planes& P, TileShared& sh, int l, int step, const ObsOut& o, bool lidar_on) {
  if (lidar_on) lidar_hits(c, P, sh, l);
  if (o.inertial) write_inertial(c, P, step, o.inertial + (size_t)P.env * TE_OBS_INERTIAL_WORDS);
  if (o.last_action) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o.last_action[(size_t)P.env * 4 + k] = P.ef(TE_E_LAST_ACTION + k);
  }
}
TE_DEV void emit_reset_obs(const te_config& c, const Planes& P, const ObsOut& o) {
  if (o.inertial) write_inertial(c, P, 0, o.inertial + (size_t)P.env * TE_OBS_INERTIAL_WORDS);
  if (o.last_action) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o.last_action[(size_t)P.env * 4 + k] = 0.0f;
  }
}

// level4 family: Env.step after advance_step (exp03_vFinal_environment.py:163-171) =
// Task.on_step_middle + compute_info + compute_observation + on_step_end, then SB3 auto-reset.
TE_DEV void level4_logic(const te_config& c, const Planes& P, TileShared& sh, int l, const float* __restrict__ actions,
                         const StepOut& o) {
  const int Pn = c.n_pursuers, D = P.D;
  {
    const float4 a = reinterpret_cast<const float4*>(actions)[P.env];
    P.ef(TE_E_LAST_ACTION + 0) = a.x; P.ef(TE_E_LAST_ACTION + 1) = a.y; P.ef(TE_E_LAST_ACTION + 2) = a.z; P.ef(TE_E_LAST_ACTION + 3) = a.w;
  }
  const int step = P.ei(TE_E_STEP) + 1;  // AGENT_STEP_BROADCAST (exp03_vFinal_environment.py:177-182)
  P.ei(TE_E_STEP) = step;
  const uint32_t episode = (uint32_t)P.ei(TE_E_EPISODE);
  const uint32_t S = armed_mask(P);  // OffsetHandler.on_middle_step: drones armed now
  P.ei(TE_E_SNAP_MASK) = (int32_t)S;

  int agent_shots = 0, ally_shots = 0, exploded = 0, pursuer_suicided = 0, agent_suicided = 0;
  // process_shoot_range_invaders (exp03_vFinal_task.py:392-413)
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = closest_invader_in_range(c, P, S, p, c.shoot_range);
    if (tgt < 0) continue;
    int mun = P.di(TE_D_MUNITION, p), lf = P.di(TE_D_LAST_FIRED, p);
    if (!(gun_available(c, mun, lf, step) && mun > 0)) continue;  // Gun.can_fire (gun.py:77-81)
    P.di(TE_D_MUNITION, p) = mun - 1;
    P.di(TE_D_LAST_FIRED, p) = step;
    U4 r = env_rng(c, P.env, RNG_HIT, (uint32_t)p, 0, episode, (uint32_t)step);
    if (u01(r.x) < c.hit_prob) {  // gun.py:94; entities_manager.shoot_by_ids (:238-248)
      disarm(P, tgt);
      if (p == 0) agent_shots += 1; else ally_shots += 1;
    }
  }
  // process_explosion_range_invaders (:359-390) on the same (stale) distances
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = closest_invader_in_range(c, P, S, p, c.explosion_range);
    if (tgt < 0) continue;
    disarm(P, p);
    disarm(P, tgt);
    int mun = P.di(TE_D_MUNITION, p);
    if (mun == 0 && p == 0) agent_suicided += 1;
    else if (mun == 0) pursuer_suicided += 1;
    else exploded += 1;
  }
  const int agent_kills = P.ei(TE_E_AGENT_KILLS) + agent_shots;
  const int allies_kills = P.ei(TE_E_ALLIES_KILLS) + ally_shots;
  const int deads = P.ei(TE_E_DEADS) + exploded;
  P.ei(TE_E_AGENT_KILLS) = agent_kills; P.ei(TE_E_ALLIES_KILLS) = allies_kills; P.ei(TE_E_DEADS) = deads;
  // process_invaders_in_origin (:656-659)
  for (int j = Pn; j < D; ++j)
    if (((S >> j) & 1u) && norm(obs_pos(P, j)) < c.origin_range) disarm(P, j);

  // compute_reward (:423-515)
  float reward;
  const V3 apos = obs_pos(P, 0);
  {
    float g[3];
    gun_state(c, P.di(TE_D_MUNITION, 0), P.di(TE_D_LAST_FIRED, 0), step, max_munition_of(c, 0), g);
    const float dist_origin = norm(apos);
    // identify_closest_ally / identify_closest_invader (offsets_handler.py:167-190,256-281)
    int ally = -1;
    if ((S & 1u) && __popc(S & ((1u << Pn) - 1u)) > 1) ally = closest_in(P, S, 0, Pn, apos, 0);
    const int from = ally < 0 ? 0 : ally;
    int target = ((S >> from) & 1u) ? closest_in(P, S, Pn, D, obs_pos(P, from), -1) : -1;
    V3 tp = target >= 0 ? obs_pos(P, target) : V3{0, 0, 0};
    const float cur = dist(apos, tp);
    const bool ready = g[2] == 1.0f || g[0] == 0.0f;
    float score, bonus = 0.0f, penalty = 0.0f;
    const float last = P.ef(TE_E_LAST_DIST);
    if (0.01f < last - cur && ready)
      bonus += c.approach_bonus_gain * norm(V3{P.df(TE_D_OBS_VEL, 0), P.df(TE_D_OBS_VEL + 1, 0), P.df(TE_D_OBS_VEL + 2, 0)});
    P.ef(TE_E_LAST_DIST) = cur;
    score = ready ? -cur : cur * (2.0f * g[1] - 1.0f);
    if (agent_shots > 0 || agent_suicided > 0) bonus += (float)(agent_shots + agent_suicided) * 1000.0f;
    if (ally_shots > 0 || pursuer_suicided > 0) bonus += 0.5f * (float)(ally_shots + pursuer_suicided) * 1000.0f;
    else if (exploded > 0) penalty += 1000.0f * (float)exploded;
    if (apos.z < -5.0f) penalty += (-5.0f - apos.z) * 1000.0f;
    bool outside = false;
    for (int p = 0; p < Pn; ++p)
      if (((S >> p) & 1u) && norm(obs_pos(P, p)) > c.dome_radius) outside = true;
    if (outside) penalty += 1000.0f;
    if (dist_origin > c.born_radius - 2.0f) penalty += dist_origin - c.born_radius - 2.0f;  // literal
    reward = score + bonus - penalty;
  }
  // increment_max_step (:150-153)
  int max_step = P.ei(TE_E_MAX_STEP);
  if (agent_shots + ally_shots > 0) { max_step += c.step_increment; P.ei(TE_E_MAX_STEP) = max_step; }
  // compute_termination (:517-569)
  int armed_invaders = 0, armed_pursuers = 0;
  for (int j = Pn; j < D; ++j) armed_invaders += P.di(TE_D_ARMED, j) ? 1 : 0;
  for (int p = 0; p < Pn; ++p) armed_pursuers += P.di(TE_D_ARMED, p) ? 1 : 0;
  int round = P.ei(TE_E_ROUND);
  const bool all_rounds_over = armed_invaders == 0 && round >= c.n_rounds;
  bool term = step > max_step || all_rounds_over;
  if (!term) {
    for (int s = 0; s < D; ++s)
      if (((S >> s) & 1u) && norm(obs_pos(P, s)) > c.dome_radius) term = true;
    if (armed_pursuers == 0 || !P.di(TE_D_ARMED, 0) || apos.z < -5.99f) term = true;
  }
  // info (:571-578)
  o.reward[P.env] = reward;
  o.done[P.env] = term ? 1 : 0;
  o.info[(size_t)P.env * 4 + 0] = agent_kills; o.info[(size_t)P.env * 4 + 1] = allies_kills;
  o.info[(size_t)P.env * 4 + 2] = deads; o.info[(size_t)P.env * 4 + 3] = round;
  // observation of THIS step (terminal observation when done)
  const bool to_terminal = term && c.auto_reset;
  sh.done[l] = to_terminal ? 1 : 0;
  emit_obs(c, P, sh, l, step, to_terminal ? o.term : o.obs, to_terminal ? o.term.lidar != nullptr : true);
  // on_step_end (:321-333)
  if (!term && armed_invaders == 0 && armed_pursuers > 0) {
    round += round < c.n_rounds ? 1 : c.n_rounds;  // advance_round (:155-175)
    P.ei(TE_E_ROUND) = round;
    level4_setup_round(c, P, round, episode);
    level4_refresh_snapshot(P);
  }
  if (to_terminal) {  // SB3 VecEnv auto-reset
    level4_reset_env(c, P);
    emit_reset_obs(c, P, o.obs);
  } else {
    prepare_level4_commands(c, P);
  }
}

// stage02: L3Stage1.on_step_middle etc. (level3/components/stages.py:144-179,241-344)
TE_DEV void stage02_logic(const te_config& c, const Planes& P, TileShared& sh, int l, const float* __restrict__ actions,
                          const StepOut& o) {
  const int Pn = c.n_pursuers, D = P.D;
  {
    const float4 a = reinterpret_cast<const float4*>(actions)[P.env];
    P.ef(TE_E_LAST_ACTION + 0) = a.x; P.ef(TE_E_LAST_ACTION + 1) = a.y; P.ef(TE_E_LAST_ACTION + 2) = a.z; P.ef(TE_E_LAST_ACTION + 3) = a.w;
  }
  const int step = P.ei(TE_E_STEP) + 1;
  P.ei(TE_E_STEP) = step;
  const uint32_t episode = (uint32_t)P.ei(TE_E_EPISODE);
  const uint32_t S = armed_mask(P);
  P.ei(TE_E_SNAP_MASK) = (int32_t)S;
  int shots = 0, exploded = 0;
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = closest_invader_in_range(c, P, S, p, c.shoot_range);
    if (tgt < 0) continue;
    int mun = P.di(TE_D_MUNITION, p), lf = P.di(TE_D_LAST_FIRED, p);
    // shoot_by_ids with the suicide rule (level3/components/quadcopter_manager.py:155-171)
    if (mun == 0) { disarm(P, tgt); shots += 1; continue; }
    if (!gun_available(c, mun, lf, step)) continue;
    P.di(TE_D_MUNITION, p) = mun - 1;
    P.di(TE_D_LAST_FIRED, p) = step;
    U4 r = env_rng(c, P.env, RNG_HIT, (uint32_t)p, 0, episode, (uint32_t)step);
    if (u01(r.x) < c.hit_prob) { disarm(P, tgt); shots += 1; }
  }
  for (int p = 0; p < Pn; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = closest_invader_in_range(c, P, S, p, c.explosion_range);
    if (tgt < 0) continue;
    disarm(P, p); disarm(P, tgt); exploded += 1;
  }
  const int kills = P.ei(TE_E_AGENT_KILLS) + shots, deads = P.ei(TE_E_DEADS) + exploded;
  P.ei(TE_E_AGENT_KILLS) = kills; P.ei(TE_E_DEADS) = deads;
  float g[3];
  gun_state(c, P.di(TE_D_MUNITION, 0), P.di(TE_D_LAST_FIRED, 0), step, max_munition_of(c, 0), g);
  const float cur = stage02_agent_min_distance(c, P, S);
  const float last = P.ef(TE_E_PREV_SNAP_MIN);
  float score, bonus = 0.0f, penalty = 0.0f;
  if (g[2] == 1.0f) score = -cur;
  else if (g[0] == 0.0f) score = -cur;
  else score = cur * (2.0f * g[1] - 1.0f);
  if (0.01f < last - cur && (g[2] == 1.0f || g[0] == 0.0f))
    bonus += c.approach_bonus_gain * norm(V3{P.df(TE_D_OBS_VEL, 0), P.df(TE_D_OBS_VEL + 1, 0), P.df(TE_D_OBS_VEL + 2, 0)});
  bonus += 1000.0f * (float)shots;
  penalty += 1000.0f * (float)exploded;
  int outside_p = 0, outside_i = 0;
  for (int s = 0; s < D; ++s)
    if (((S >> s) & 1u) && norm(obs_pos(P, s)) > c.dome_radius) { if (s < Pn) outside_p += 1; else outside_i += 1; }
  if (outside_p > 0) penalty += 1000.0f;
  const float reward = score + bonus - penalty;
  int armed_pursuers = 0;
  for (int p = 0; p < Pn; ++p) armed_pursuers += P.di(TE_D_ARMED, p) ? 1 : 0;
  const bool term = step > P.ei(TE_E_MAX_STEP) || outside_p > 0 || outside_i > 0 || armed_pursuers < Pn;
  o.reward[P.env] = reward;
  o.done[P.env] = term ? 1 : 0;
  o.info[(size_t)P.env * 4 + 0] = kills; o.info[(size_t)P.env * 4 + 1] = 0;
  o.info[(size_t)P.env * 4 + 2] = deads; o.info[(size_t)P.env * 4 + 3] = 0;
  // observation before the respawn: a drone armed after the step broadcast has no Delta=1 snapshot
  // yet (lidar_buffer.py:443-447) and is invisible this step
  const bool to_terminal = term && c.auto_reset;
  sh.done[l] = to_terminal ? 1 : 0;
  emit_obs(c, P, sh, l, step, to_terminal ? o.term : o.obs, to_terminal ? o.term.lidar != nullptr : true);
  // respawn killed invaders (stages.py:167-174)
  for (int j = Pn; j < D; ++j)
    if (!P.di(TE_D_ARMED, j)) respawn_armed(c, P, j, stage02_invader_position(c, P, j, episode, (uint32_t)step));
  P.ef(TE_E_PREV_SNAP_MIN) = cur; P.ef(TE_E_LAST_DIST) = cur;  // on_step_end: last_offsets = current_offsets
  if (to_terminal) { stage02_reset_env(c, P); emit_reset_obs(c, P, o.obs); }
}

// stage01: PyflytL2EnviromentModifiedV2.step after the sim loop (pyflyt_level2_environment_modified_v2.py:137-145)
TE_DEV void stage01_logic(const te_config& c, const Planes& P, TileShared& sh, int l, const float* __restrict__ actions,
                          const StepOut& o) {
  {
    const float4 a = reinterpret_cast<const float4*>(actions)[P.env];
    P.ef(TE_E_LAST_ACTION + 0) = a.x; P.ef(TE_E_LAST_ACTION + 1) = a.y; P.ef(TE_E_LAST_ACTION + 2) = a.z; P.ef(TE_E_LAST_ACTION + 3) = a.w;
  }
  const int step = P.ei(TE_E_STEP) + 1;  // step_calls
  P.ei(TE_E_STEP) = step;
  const uint32_t episode = (uint32_t)P.ei(TE_E_EPISODE);
  const V3 pp = obs_pos(P, 0), pi = obs_pos(P, 2);
  const float d = dist(pi, pp);
  float bonus = 0.0f, penalty = 0.0f;
  if (d < P.ef(TE_E_LAST_DIST))
    bonus += c.approach_bonus_gain * norm(V3{P.df(TE_D_OBS_VEL, 0), P.df(TE_D_OBS_VEL + 1, 0), P.df(TE_D_OBS_VEL + 2, 0)});
  if (d < c.catch_distance) bonus += 1000.0f;
  if (d > c.dome_radius) penalty += 1000.0f;
  const float reward = -d + bonus - penalty;
  const bool term = step > P.ei(TE_E_MAX_STEP) || norm(pp) > c.dome_radius || norm(pi) > c.dome_radius;
  int kills = P.ei(TE_E_AGENT_KILLS);
  const bool to_terminal = term && c.auto_reset;
  sh.done[l] = to_terminal ? 1 : 0;
  emit_obs(c, P, sh, l, step, to_terminal ? o.term : o.obs, to_terminal ? o.term.lidar != nullptr : true);
  if (d < c.catch_distance) {  // replace_invader_if_close (:147-154)
    stage01_replace_invader(c, P, stage01_cube(c, P, RNG_RESPAWN, 2, episode, (uint32_t)step), episode, (uint32_t)step);
    kills += 1;
    P.ei(TE_E_AGENT_KILLS) = kills;
  }
  P.ef(TE_E_LAST_DIST) = dist(obs_pos(P, 2), obs_pos(P, 0));  // update_last_distance (:219-223)
  o.reward[P.env] = reward;
  o.done[P.env] = term ? 1 : 0;
  o.info[(size_t)P.env * 4 + 0] = kills; o.info[(size_t)P.env * 4 + 1] = 0;
  o.info[(size_t)P.env * 4 + 2] = 0; o.info[(size_t)P.env * 4 + 3] = 0;
  if (to_terminal) { stage01_reset_env(c, P); emit_reset_obs(c, P, o.obs); }
}

template <int FAMILY>
__global__ __launch_bounds__(256) void engage_observe_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  __shared__ TileShared sh;
  const int env0 = blockIdx.x * kEnvsPerBlock;
  const int nvalid = min(kEnvsPerBlock, p.N - env0);
  clear_maps(sh);
  __syncthreads();
  if (threadIdx.x < kEnvsPerBlock && (int)threadIdx.x < nvalid) {
    Planes P{p.dstate, p.estate, p.D, p.Npad, env0 + (int)threadIdx.x};
    if (FAMILY == FAM_STAGE01) stage01_logic(p.cfg, P, sh, threadIdx.x, actions, o);
    else if (FAMILY == FAM_STAGE02) stage02_logic(p.cfg, P, sh, threadIdx.x, actions, o);
    else level4_logic(p.cfg, P, sh, threadIdx.x, actions, o);
  }
  __syncthreads();
  if (o.obs.lidar) write_tiles(sh, o.obs.lidar, env0, nvalid, true);
  if (o.term.lidar) write_terminal_tiles(sh, o.term.lidar, env0, nvalid);
}

// te_observe: current observation without stepping
__global__ __launch_bounds__(256) void observe_kernel(Params p, ObsOut o) {
  __shared__ TileShared sh;
  const int env0 = blockIdx.x * kEnvsPerBlock;
  const int nvalid = min(kEnvsPerBlock, p.N - env0);
  clear_maps(sh);
  __syncthreads();
  if (threadIdx.x < kEnvsPerBlock && (int)threadIdx.x < nvalid) {
    Planes P{p.dstate, p.estate, p.D, p.Npad, env0 + (int)threadIdx.x};
    const int step = P.ei(TE_E_STEP);
    // right after a reset the Delta=1 snapshot does not exist yet: empty sphere
    emit_obs(p.cfg, P, sh, threadIdx.x, step, o, step != 0 && o.lidar != nullptr);
  }
  __syncthreads();
  if (o.lidar) write_tiles(sh, o.lidar, env0, nvalid, false);
}

// ============================================================================================
// state blob <-> planes, synthetic actions
// ============================================================================================
__global__ void blob_to_planes(Params p, const uint32_t* __restrict__ blob) {
  const size_t ND = (size_t)p.N * p.D;
  const size_t drone_words = ND * TE_DRONE_WORDS, total = drone_words + (size_t)p.N * TE_ENV_WORDS;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (i < drone_words) {
      size_t rec = i / TE_DRONE_WORDS; int w = (int)(i - rec * TE_DRONE_WORDS);
      int e = (int)(rec / p.D), d = (int)(rec - (size_t)e * p.D);
      p.dstate[((size_t)w * p.D + d) * p.Npad + e] = blob[i];
    } else {
      size_t j = i - drone_words; int e = (int)(j / TE_ENV_WORDS), w = (int)(j - (size_t)e * TE_ENV_WORDS);
      p.estate[(size_t)w * p.Npad + e] = blob[i];
    }
  }
}
__global__ void planes_to_blob(Params p, uint32_t* __restrict__ blob) {
  const size_t ND = (size_t)p.N * p.D;
  const size_t drone_words = ND * TE_DRONE_WORDS, total = drone_words + (size_t)p.N * TE_ENV_WORDS;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (i < drone_words) {
      size_t rec = i / TE_DRONE_WORDS; int w = (int)(i - rec * TE_DRONE_WORDS);
      int e = (int)(rec / p.D), d = (int)(rec - (size_t)e * p.D);
      blob[i] = p.dstate[((size_t)w * p.D + d) * p.Npad + e];
    } else {
      size_t j = i - drone_words; int e = (int)(j / TE_ENV_WORDS), w = (int)(j - (size_t)e * TE_ENV_WORDS);
      blob[i] = p.estate[(size_t)w * p.Npad + e];
    }
  }
}
__global__ void init_planes(Params p) {  // identity attitude everywhere
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)p.D * p.Npad; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<float*>(p.dstate)[(size_t)(TE_D_QUAT + 3) * p.D * p.Npad + i] = 1.0f;
}
// dir ~ U(-1,1)^3, mag ~ U(0,1) (apps/threatengage_runner/interactive/analyse.py:55-59)
__global__ void random_actions_kernel(Params p, float* __restrict__ actions, uint64_t seed, uint64_t step_index) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= p.N) return;
  uint64_t g = (uint64_t)p.cfg.env_index_base + (uint64_t)env;
  U4 r = philox4x32_10((uint32_t)g, RNG_ACTION | ((uint32_t)(g >> 32) << 24), (uint32_t)step_index, (uint32_t)(step_index >> 32),
                       (uint32_t)seed, (uint32_t)(seed >> 32));
  reinterpret_cast<float4*>(actions)[env] = make_float4(-1.0f + 2.0f * u01(r.x), -1.0f + 2.0f * u01(r.y), -1.0f + 2.0f * u01(r.z), u01(r.w));
}

}  // namespace te

// ==============================================================================================
// C ABI
// ==============================================================================================
using namespace te;

struct te_env {
  Params p;
  int device;
  int family;
  // profiling (te_profile_begin / te_profile_end)
  std::vector<hipEvent_t> events;
  int prof_cap = 0, prof_used = 0;
};

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define TE_HIP(x)                                                                                      \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_));                  \
  } while (0)

struct DeviceGuard {
  int prev = -1; bool ok;
  explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess); if (prev == dev) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int family_of(int task) {
  return task == TE_TASK_STAGE01 ? FAM_STAGE01 : (task == TE_TASK_STAGE02 ? FAM_STAGE02 : FAM_LEVEL4);
}

template <typename F>
static int launch_by_family(int family, F&& f) {
  switch (family) {
    case FAM_STAGE01: f(std::integral_constant<int, FAM_STAGE01>{}); break;
    case FAM_STAGE02: f(std::integral_constant<int, FAM_STAGE02>{}); break;
    default: f(std::integral_constant<int, FAM_LEVEL4>{}); break;
  }
  return 0;
}

extern "C" {

__attribute__((visibility("default"))) const char* te_last_error(void) { return g_err.c_str(); }

__attribute__((visibility("default"))) int te_create(const te_config* cfg, int32_t device_id, te_env** out) {
  if (!cfg || !out) return fail("te_create: null argument");
  if (cfg->struct_size != sizeof(te_config)) return fail("te_create: te_config.struct_size mismatch (ABI skew)");
  const int D = cfg->n_pursuers + cfg->n_invaders;
  if (cfg->n_envs < 1) return fail("te_create: n_envs < 1");
  if (cfg->n_pursuers < 1 || cfg->n_invaders < 1 || D > kMaxD) return fail("te_create: need 1 <= P, 1 <= I, P + I <= 32");
  if (cfg->substeps < 1 || cfg->substeps > 255) return fail("te_create: substeps out of range");
  if (cfg->task < TE_TASK_STAGE01 || cfg->task > TE_TASK_EXP04) return fail("te_create: unknown task");
  if (cfg->task == TE_TASK_STAGE01 && !(cfg->n_pursuers == 2 && cfg->n_invaders == 1)) return fail("te_create: stage01 is 2 pursuers + 1 invader");
  if (cfg->lidar_radius <= 0.0f || cfg->dome_radius <= 0.0f || cfg->max_speed <= 0.0f) return fail("te_create: radii / max_speed must be positive");
  int ndev = 0;
  TE_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail("te_create: no HIP device visible; this library has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return fail("te_create: bad device_id");
  DeviceGuard guard(device_id);
  if (!guard.ok) return fail("te_create: hipSetDevice failed");
  te_env* e = new (std::nothrow) te_env();
  if (!e) return fail("te_create: out of host memory");
  e->device = device_id;
  e->family = family_of(cfg->task);
  e->p.cfg = *cfg;
  e->p.N = cfg->n_envs; e->p.D = D; e->p.Npad = (cfg->n_envs + 63) / 64 * 64;
  const size_t dwords = (size_t)(TE_DRONE_WORDS + TE_X_WORDS) * D * e->p.Npad, ewords = (size_t)TE_ENV_WORDS * e->p.Npad;
  if (hipMalloc(&e->p.dstate, dwords * 4) != hipSuccess || hipMalloc(&e->p.estate, ewords * 4) != hipSuccess) {
    if (e->p.dstate) (void)hipFree(e->p.dstate);
    delete e;
    return fail("te_create: hipMalloc failed");
  }
  TE_HIP(hipMemsetAsync(e->p.dstate, 0, dwords * 4, nullptr));
  TE_HIP(hipMemsetAsync(e->p.estate, 0, ewords * 4, nullptr));
  hipLaunchKernelGGL(init_planes, dim3(256), dim3(256), 0, nullptr, e->p);
  const int blocks = (e->p.N + 255) / 256;
  launch_by_family(e->family, [&](auto fam) {
    hipLaunchKernelGGL((reset_kernel<decltype(fam)::value>), dim3(blocks), dim3(256), 0, nullptr, e->p, (const uint8_t*)nullptr);
  });
  TE_HIP(hipGetLastError());
  TE_HIP(hipStreamSynchronize(nullptr));
  *out = e;
  return 0;
}

__attribute__((visibility("default"))) void te_destroy(te_env* e) {
  if (!e) return;
  DeviceGuard guard(e->device);
  for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
  (void)hipFree(e->p.dstate);
  (void)hipFree(e->p.estate);
  delete e;
}

__attribute__((visibility("default"))) int te_reset(te_env* e, const uint8_t* env_mask, void* stream) {
  if (!e) return fail("te_reset: null env");
  DeviceGuard guard(e->device);
  const int blocks = (e->p.N + 255) / 256;
  launch_by_family(e->family, [&](auto fam) {
    hipLaunchKernelGGL((reset_kernel<decltype(fam)::value>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, e->p, env_mask);
  });
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_observe(te_env* e, float* obs_lidar, float* obs_inertial, float* obs_last_action, void* stream) {
  if (!e) return fail("te_observe: null env");
  if (obs_lidar && ((uintptr_t)obs_lidar & 15)) return fail("te_observe: obs_lidar must be 16-byte aligned");
  DeviceGuard guard(e->device);
  const int blocks = (e->p.N + kEnvsPerBlock - 1) / kEnvsPerBlock;
  hipLaunchKernelGGL(observe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, e->p, ObsOut{obs_lidar, obs_inertial, obs_last_action});
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_step(te_env* e, const float* actions, float* obs_lidar, float* obs_inertial,
                                                   float* obs_last_action, float* reward, uint8_t* done, int32_t* info,
                                                   float* terminal_lidar, float* terminal_inertial, float* terminal_last_action,
                                                   void* stream) {
  if (!e) return fail("te_step: null env");
  if (!actions || !reward || !done || !info) return fail("te_step: actions, reward, done and info are required");
  if (((uintptr_t)actions & 15) || (obs_lidar && ((uintptr_t)obs_lidar & 15)))
    return fail("te_step: actions and obs_lidar must be 16-byte aligned");
  DeviceGuard guard(e->device);
  hipStream_t st = (hipStream_t)stream;
  const Params& p = e->p;
  const bool prof = e->prof_used + 3 <= e->prof_cap;
  if (prof) TE_HIP(hipEventRecord(e->events[e->prof_used + 0], st));
  const int waves = p.D * (p.Npad >> 6);
  const int b1 = (waves + 3) / 4;
  const bool noise = p.cfg.motor_noise != 0;
  launch_by_family(e->family, [&](auto fam) {
    constexpr int F = decltype(fam)::value;
    if (noise) hipLaunchKernelGGL((substeps_kernel<F, true>), dim3(b1), dim3(256), 0, st, p, actions);
    else hipLaunchKernelGGL((substeps_kernel<F, false>), dim3(b1), dim3(256), 0, st, p, actions);
  });
  if (prof) TE_HIP(hipEventRecord(e->events[e->prof_used + 1], st));
  const int b2 = (p.N + kEnvsPerBlock - 1) / kEnvsPerBlock;
  StepOut o{reward, done, info, ObsOut{obs_lidar, obs_inertial, obs_last_action},
            ObsOut{terminal_lidar, terminal_inertial, terminal_last_action}};
  launch_by_family(e->family, [&](auto fam) {
    hipLaunchKernelGGL((engage_observe_kernel<decltype(fam)::value>), dim3(b2), dim3(256), 0, st, p, actions, o);
  });
  if (prof) { TE_HIP(hipEventRecord(e->events[e->prof_used + 2], st)); e->prof_used += 3; }
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_random_actions(te_env* e, float* actions, uint64_t seed, uint64_t step_index, void* stream) {
  if (!e || !actions) return fail("te_random_actions: null argument");
  if ((uintptr_t)actions & 15) return fail("te_random_actions: actions must be 16-byte aligned");
  DeviceGuard guard(e->device);
  hipLaunchKernelGGL(random_actions_kernel, dim3((e->p.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->p, actions, seed, step_index);
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_state_words(const te_env* e, size_t* out_words) {
  if (!e || !out_words) return fail("te_state_words: null argument");
  *out_words = (size_t)e->p.N * ((size_t)e->p.D * TE_DRONE_WORDS + TE_ENV_WORDS);
  return 0;
}

__attribute__((visibility("default"))) int te_get_state(te_env* e, void* dst_device, size_t words, void* stream) {
  size_t need = 0;
  if (te_state_words(e, &need)) return 1;
  if (!dst_device || words != need) return fail("te_get_state: buffer must hold exactly te_state_words() words");
  DeviceGuard guard(e->device);
  hipLaunchKernelGGL(planes_to_blob, dim3(1024), dim3(256), 0, (hipStream_t)stream, e->p, (uint32_t*)dst_device);
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_set_state(te_env* e, const void* src_device, size_t words, void* stream) {
  size_t need = 0;
  if (te_state_words(e, &need)) return 1;
  if (!src_device || words != need) return fail("te_set_state: buffer must hold exactly te_state_words() words");
  DeviceGuard guard(e->device);
  hipLaunchKernelGGL(blob_to_planes, dim3(1024), dim3(256), 0, (hipStream_t)stream, e->p, (const uint32_t*)src_device);
  if (e->family == FAM_LEVEL4)
    hipLaunchKernelGGL(prepare_commands_kernel, dim3((e->p.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->p);
  TE_HIP(hipGetLastError());
  return 0;
}

// Kernel timing with HIP events on the stream te_step launches on.
__attribute__((visibility("default"))) int te_profile_begin(te_env* e, int32_t max_steps) {
  if (!e || max_steps < 1) return fail("te_profile_begin: bad argument");
  DeviceGuard guard(e->device);
  while ((int)e->events.size() < 3 * max_steps) {
    hipEvent_t ev;
    TE_HIP(hipEventCreate(&ev));
    e->events.push_back(ev);
  }
  e->prof_cap = 3 * max_steps;
  e->prof_used = 0;
  return 0;
}
// Averages over the steps recorded since te_profile_begin; blocks until they have finished.
__attribute__((visibility("default"))) int te_profile_end(te_env* e, float* substeps_ms, float* engage_observe_ms, int32_t* n_steps) {
  if (!e) return fail("te_profile_end: null env");
  DeviceGuard guard(e->device);
  const int n = e->prof_used / 3;
  double a = 0, b = 0;
  if (n > 0) TE_HIP(hipEventSynchronize(e->events[e->prof_used - 1]));
  for (int i = 0; i < n; ++i) {
    float m1 = 0, m2 = 0;
    TE_HIP(hipEventElapsedTime(&m1, e->events[3 * i], e->events[3 * i + 1]));
    TE_HIP(hipEventElapsedTime(&m2, e->events[3 * i + 1], e->events[3 * i + 2]));
    a += m1; b += m2;
  }
  if (substeps_ms) *substeps_ms = n ? (float)(a / n) : 0.0f;
  if (engage_observe_ms) *engage_observe_ms = n ? (float)(b / n) : 0.0f;
  if (n_steps) *n_steps = n;
  e->prof_cap = 0; e->prof_used = 0;
  return 0;
}

}  // extern "C"
