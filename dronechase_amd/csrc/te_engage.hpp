// te_engage.hpp — the level4-family engage/observe kernel with the environment in REGISTERS.
//
// engage_observe_kernel (te_env.hip) runs a 256-thread block per 64 envs through barrier-separated phases over an LDS
// copy of the block's state; its span is the critical path of one block (stage 6.5 + precompute 3.5 + logic 7.4 + spawn/rows
// 5-12 + patch 0.5-5 us) and 60 % of its wave-cycles wait at s_waitcnt (profiles/r01_f_pmc_waits.txt).  At 65 536 envs there
// are only 1 024 chunks for 1 024 SIMDs, so nothing is gained by the block's extra waves except the phases' parallelism, and
// that is bought with six barriers and an LDS round trip for every operand.
//
// engage_kernel<PM, IM> instead gives every env ONE lane for the whole step and every chunk of 64 envs ONE wave (a
// single-wave workgroup: no barrier ever waits for another wave):
//   * every word the logic reads (IMU positions, armed flags, the pursuers' guns, the agent's IMU, the env record, the
//     action) is requested up front with independent coalesced loads (planes are env-fastest: one load = one 256-byte
//     line per wave) and lands in registers: ONE memory latency, then straight-line arithmetic.  The slot loops run over the
//     compile-time capacity (PM pursuers, PM + IM slots) with the run-time counts as predicates, so the per-slot arrays
//     stay in VGPRs (a run-time trip count would send them to scratch);
//   * engagement, reward, termination, info, closer-wins LIDAR resolution, wave advance / auto-reset decision: the
//     reference's order (exp03_vFinal_task.py:285-413,423-578), on bit masks;
//   * the rare heavy parts are done by the WHOLE wave for the env that needs them: a new round or an auto-reset respawns up
//     to D drones (Philox + trigonometry + ~40 stores each): lane k takes slot k; a terminal LIDAR tile (1 014 floats of
//     ones) is streamed by all 64 lanes;
//   * the allies' commands and the flight plan of the next sub-step launch come from the same registers.
// Kernel span = one wave's straight-line path (~2 000 VALU instructions + one load round trip) instead of a block's phase chain.
//
// Covers the level4 task family (exp02/03/04/05, evaluation, level5's engage step) for P <= PM and P + I <= PM + IM; other
// shapes and the stage01 / stage02 families keep engage_observe_kernel.  TE_ENGAGE=lds selects the LDS kernel for A/B runs.
//
// Reference citations are file:line under the reference's src/ tree.
#pragma once
#include "te_logic.hpp"
#include "te_stacked.hpp"
#include <type_traits>

namespace te {

#if defined(TE_DEBUG_STAMPS) && !defined(TE_NO_ESTAMP)
// phase stamps of every workgroup (diagnostic builds only; tools/engage_stamps.py): idx 0..15 of the block's record
#define TE_ESTAMP(idx, wait)                                                                                   \
  do {                                                                                                         \
    if (wait) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
    if (p.dbg && threadIdx.x == 0) p.dbg[64 + blockIdx.x * 16 + (idx)] = __builtin_amdgcn_s_memrealtime();     \
  } while (0)
#else
#define TE_ESTAMP(idx, wait) do {} while (0)
#endif

template <class V> TE_DEV void plan_slot_l4(uint16_t* __restrict__ items, int dense_min, int lane, int s, bool a, uint64_t& dense, int& n, uint64_t& live) {
  const unsigned long long b = __ballot(a);
  const int cnt = __popcll(b);
  if (cnt == 0) return;
  live |= (uint64_t)1 << s;
  if (s == 0 || cnt >= dense_min || n + cnt > kMixedCap) { dense |= (uint64_t)1 << s; return; }
  if (a) items[n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = (uint16_t)(lane | (s << 8));
  n += cnt;
}

// ---- exact arithmetic ------------------------------------------------------------------------------------------------------------
// engage_kernel (one wave per chunk) and engage_slots_kernel (te_engage_slots.hpp: one wave per (chunk, slot)) must produce the same bits:
// a shard stepped by one and the whole env range stepped by the other are compared bitwise.  With -ffp-contract=fast the compiler picks
// which multiply of a sum it fuses from the USE COUNTS around the expression, i.e. from the kernel the function was inlined into (first
// slots build: 10 % of the LIDAR ranges one ulp apart).  The float arithmetic both kernels share is therefore written with contraction OFF
// and every fused multiply-add spelled out (TE_EXACT, xfma, x_asin, x_atan2: te_device.hpp).
// |a| and |a - b| on the native square root (1 ulp) instead of the correctly rounded sqrtf (~10 instructions each, ~45 calls)
TE_DEV float fnorm(V3 a) { TE_EXACT return fsqrt(xfma(a.z, a.z, xfma(a.y, a.y, a.x * a.x))); }
TE_DEV float fdist(V3 a, V3 b) { TE_EXACT return fnorm(V3{a.x - b.x, a.y - b.y, a.z - b.z}); }
// the agent's inverse attitude from its IMU euler angles: getQuaternionFromEuler (te_device.hpp: quat_of_euler), conjugate / |q|^2,
// btMatrix3x3::setRotation (te_device.hpp: rotation) — the same formulas with the contraction fixed
TE_DEV M3 x_inverse_attitude(float roll, float pitch, float yaw) {
  TE_EXACT
  float sr, cr, sp, cp, sy, cy;
  x_sincos(0.5f * roll, sr, cr); x_sincos(0.5f * pitch, sp, cp); x_sincos(0.5f * yaw, sy, cy);
  const float a = sr * cp, b = cr * sp, d = cr * cp, e = sr * sp;
  float qx = xfma(a, cy, -(b * sy)), qy = xfma(b, cy, a * sy), qz = xfma(d, sy, -(e * cy)), qw = xfma(d, cy, e * sy);
  const float inv = 1.0f / sqrtf(xfma(qw, qw, xfma(qz, qz, xfma(qy, qy, qx * qx))));
  qx *= inv; qy *= inv; qz *= inv; qw *= inv;
  const float n2 = xfma(qw, qw, xfma(qz, qz, xfma(qy, qy, qx * qx)));
  const float ix = -qx / n2, iy = -qy / n2, iz = -qz / n2, iw = qw / n2;
  const float dd = xfma(iw, iw, xfma(iz, iz, xfma(iy, iy, ix * ix)));
  const float s2 = 2.0f * rcp(dd);
  const float xs = ix * s2, ys = iy * s2, zs = iz * s2;
  const float wx = iw * xs, wy = iw * ys, wz = iw * zs;
  const float xx = ix * xs, xy = ix * ys, xz = ix * zs;
  const float yy = iy * ys, yz = iy * zs, zz = iz * zs;
  return M3{1.0f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0f - (xx + yy)};
}
TE_DEV V3 x_mul(const M3& R, V3 v) {
  TE_EXACT
  return V3{xfma(R.m02, v.z, xfma(R.m01, v.y, R.m00 * v.x)), xfma(R.m12, v.z, xfma(R.m11, v.y, R.m10 * v.x)), xfma(R.m22, v.z, xfma(R.m21, v.y, R.m20 * v.x))};
}
// lidar_cell (te_logic.hpp) on the native sqrt / rcp and the polynomial asin / atan2 of the sub-step loop (1e-7 abs): a tenth of
// libm's acosf + atan2f instructions, which were a third of this kernel's straight-line path
TE_DEV void lidar_cell_fast(const te_config& c, V3 local, int& cell, float& rhat) {
  TE_EXACT
  const float r2 = xfma(local.z, local.z, xfma(local.y, local.y, local.x * local.x));
  float theta = 0.0f, phi = 0.0f, r = 0.0f;
  if (r2 != 0.0f) {
    const float inv = rsq(r2);
    r = r2 * inv;
    theta = 0.5f * kPi - x_asin(clampf(local.z * inv, -1.0f, 1.0f));
    phi = x_atan2(local.y, local.x);
  }
  rhat = clampf(r * rcp(c.lidar_radius), 0.0f, 1.0f);
  const int ti = min(max((int)(theta * (1.0f / kPi) * (float)TE_LIDAR_NTHETA), 0), TE_LIDAR_NTHETA - 1);
  const int pi = min(max((int)((phi + kPi) * (0.5f / kPi) * (float)TE_LIDAR_NPHI), 0), TE_LIDAR_NPHI - 1);
  cell = ti * TE_LIDAR_NPHI + pi;
}

// Task.setup_round / Task.on_reset for ONE slot (te_logic.hpp: level4_spawn_slot = disarm [+ disarm + replace + IMU + arm]), with
// every word written ONCE and nothing written for a slot that is dead and stays dead: 40 stores for a respawned drone, 15 for one
// that is only disarmed, 1 for the others (the generic sequence writes 61 / 15 / 15).  A wave that respawns an env carries these
// stores on its critical path.  `was_armed`: the slot's ARMED flag before the call.  Returns whether the slot ends up armed, at `where`.
TE_DEV bool spawn_slot_at(const te_config& c, const GView& v, int s, int round, uint32_t episode, bool reset, bool was_armed, V3& where) {
  const int Pn = c.n_pursuers;
  const bool invader = s >= Pn;
  const bool respawn = invader ? (s - Pn < invaders_in_round(c, round)) : reset;
  if (respawn) {
    const U4 r = invader ? env_rng(c, v.env, RNG_SPAWN_INVADER, (uint32_t)s, 0, episode, (uint32_t)round)
                         : env_rng(c, v.env, RNG_SPAWN_PURSUER, (uint32_t)s, 0, episode, 0);
    where = level4_position(c, invader ? c.born_radius : c.pursuer_spawn_radius, u01(r.x), u01(r.y));
    const float w3[3] = {where.x, where.y, where.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // replace (quadcopter.py:433-439) + the IMU read of arm() (:445-459) + disarm's zeroes (:461-478)
      v.sf(TE_D_POS + k, s, w3[k]); v.sf(TE_D_FORMATION + k, s, w3[k]); v.sf(TE_D_OBS_POS + k, s, w3[k]);
      v.sf(TE_D_VEL + k, s, 0.0f); v.sf(TE_D_OMEGA + k, s, 0.0f);
      v.sf(TE_D_OBS_EULER + k, s, 0.0f); v.sf(TE_D_OBS_VEL + k, s, 0.0f); v.sf(TE_D_OBS_RATE + k, s, 0.0f);
      v.sf(TE_D_QUAT + k, s, 0.0f);
    }
    v.sf(TE_D_QUAT + 3, s, 1.0f);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v.sf(TE_D_THROTTLE + k, s, 0.0f); v.sf(TE_D_SETPOINT + k, s, 0.0f); }
    v.si(TE_D_ARMED, s, 1); v.si(TE_D_MUNITION, s, max_munition_of(c, s)); v.si(TE_D_LAST_FIRED, s, -c.cooldown_steps);
    if (!invader && driven_externally(c, s)) {   // Exp05_vFinal_Task.init_globals: last_action = zeros
#pragma unroll
      for (int k = 0; k < 4; ++k) v.sf(TE_D_ALLY_ACTION + k, s, 0.0f);
    }
  } else if (invader && was_armed) disarm(v, s);   // disarm_all_invaders: the others are dead already, their words are zero
  if (invader || reset) v.si(TE_D_NAV_STATE, s, TE_NAV_WAIT);
  return respawn;
}

// buffer-addressed access to the state planes of this lane's env (see engage_kernel)
struct EnvIO {
  __amdgpu_buffer_rsrc_t rd, re;
  int voff; uint32_t plane; uint32_t D;
  TE_DEV EnvIO(const Params& p, int env)
      : rd(__builtin_amdgcn_make_buffer_rsrc(p.dstate, 0, (int)((uint32_t)(TE_DRONE_WORDS + TE_X_WORDS) * (uint32_t)p.D * (uint32_t)p.Npad * 4u), 0x00020000)),
        re(__builtin_amdgcn_make_buffer_rsrc(p.estate, 0, (int)((uint32_t)TE_ENV_WORDS * (uint32_t)p.Npad * 4u), 0x00020000)),
        voff(env * 4), plane((uint32_t)p.Npad * 4u), D((uint32_t)p.D) {}
  TE_DEV uint32_t ld(int w, int s) const { return __builtin_amdgcn_raw_buffer_load_b32(rd, voff, (int)(((uint32_t)w * D + (uint32_t)s) * plane), 0); }
  TE_DEV float ldf(int w, int s) const { return __uint_as_float(ld(w, s)); }
  TE_DEV uint32_t le(int w) const { return __builtin_amdgcn_raw_buffer_load_b32(re, voff, (int)((uint32_t)w * plane), 0); }
  TE_DEV void st(int w, int s, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b32(v, rd, voff, (int)(((uint32_t)w * D + (uint32_t)s) * plane), 0); }
  TE_DEV void stf(int w, int s, float v) const { st(w, s, __float_as_uint(v)); }
  TE_DEV void stv(int w, int slot, uint32_t v) const {  // per-lane slot
    __builtin_amdgcn_raw_buffer_store_b32(v, rd, (int)(((uint32_t)slot * (plane >> 2)) * 4u) + voff, (int)((uint32_t)w * D * plane), 0);
  }
  TE_DEV void ste(int w, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b32(v, re, voff, (int)((uint32_t)w * plane), 0); }
  TE_DEV void stef(int w, float v) const { ste(w, __float_as_uint(v)); }
  // Quadcopter.disarm (quadcopter.py:461-478) of a per-lane slot
  TE_DEV void disarm(int slot) const {
    stv(TE_D_ARMED, slot, 0u);
#pragma unroll
    for (int k = 0; k < 3; ++k) { stv(TE_D_VEL + k, slot, 0u); stv(TE_D_OMEGA + k, slot, 0u); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { stv(TE_D_THROTTLE + k, slot, 0u); stv(TE_D_SETPOINT + k, slot, 0u); }
  }
  // disarm + replace + IMU + arm of a per-lane slot that ends up armed at `w` (respawn_armed, te_logic.hpp), every word once
  TE_DEV void respawn(const te_config& c, int slot, V3 w) const {
    const float w3[3] = {w.x, w.y, w.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      stv(TE_D_POS + k, slot, __float_as_uint(w3[k])); stv(TE_D_FORMATION + k, slot, __float_as_uint(w3[k])); stv(TE_D_OBS_POS + k, slot, __float_as_uint(w3[k]));
      stv(TE_D_VEL + k, slot, 0u); stv(TE_D_OMEGA + k, slot, 0u); stv(TE_D_OBS_EULER + k, slot, 0u); stv(TE_D_OBS_VEL + k, slot, 0u);
      stv(TE_D_OBS_RATE + k, slot, 0u); stv(TE_D_QUAT + k, slot, 0u);
    }
    stv(TE_D_QUAT + 3, slot, __float_as_uint(1.0f));
#pragma unroll
    for (int k = 0; k < 4; ++k) { stv(TE_D_THROTTLE + k, slot, 0u); stv(TE_D_SETPOINT + k, slot, 0u); }
    stv(TE_D_ARMED, slot, 1u); stv(TE_D_MUNITION, slot, (uint32_t)max_munition_of(c, slot)); stv(TE_D_LAST_FIRED, slot, (uint32_t)(-c.cooldown_steps));
  }
  // spawn_slot_at for a WAVE-UNIFORM slot (the slot waves): the same words, every store with a scalar plane offset instead of a 64-bit pointer
  // (40 stores x ~6 VALU instructions of address arithmetic were most of what a respawning wave added to its workgroup's chain)
  TE_DEV bool spawn_uniform(const te_config& c, int env, int s, int round, uint32_t episode, bool reset, bool was_armed, V3& where) const {
    const int Pn = c.n_pursuers;
    const bool invader = s >= Pn;
    const bool respawn = invader ? (s - Pn < invaders_in_round(c, round)) : reset;
    if (respawn) {
      const U4 r = invader ? env_rng(c, env, RNG_SPAWN_INVADER, (uint32_t)s, 0, episode, (uint32_t)round)
                           : env_rng(c, env, RNG_SPAWN_PURSUER, (uint32_t)s, 0, episode, 0);
      where = level4_position(c, invader ? c.born_radius : c.pursuer_spawn_radius, u01(r.x), u01(r.y));
      const float w3[3] = {where.x, where.y, where.z};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        stf(TE_D_POS + k, s, w3[k]); stf(TE_D_FORMATION + k, s, w3[k]); stf(TE_D_OBS_POS + k, s, w3[k]);
        st(TE_D_VEL + k, s, 0u); st(TE_D_OMEGA + k, s, 0u); st(TE_D_OBS_EULER + k, s, 0u); st(TE_D_OBS_VEL + k, s, 0u); st(TE_D_OBS_RATE + k, s, 0u);
        st(TE_D_QUAT + k, s, 0u);
      }
      stf(TE_D_QUAT + 3, s, 1.0f);
#pragma unroll
      for (int k = 0; k < 4; ++k) { st(TE_D_THROTTLE + k, s, 0u); st(TE_D_SETPOINT + k, s, 0u); }
      st(TE_D_ARMED, s, 1u); st(TE_D_MUNITION, s, (uint32_t)max_munition_of(c, s)); st(TE_D_LAST_FIRED, s, (uint32_t)(-c.cooldown_steps));
      if (!invader && driven_externally(c, s)) {   // Exp05_vFinal_Task.init_globals: last_action = zeros
#pragma unroll
        for (int k = 0; k < 4; ++k) st(TE_D_ALLY_ACTION + k, s, 0u);
      }
    } else if (invader && was_armed) {   // disarm_all_invaders: the others are dead already, their words are zero
      st(TE_D_ARMED, s, 0u);
#pragma unroll
      for (int k = 0; k < 3; ++k) { st(TE_D_VEL + k, s, 0u); st(TE_D_OMEGA + k, s, 0u); }
#pragma unroll
      for (int k = 0; k < 4; ++k) { st(TE_D_THROTTLE + k, s, 0u); st(TE_D_SETPOINT + k, s, 0u); }
    }
    if (invader || reset) st(TE_D_NAV_STATE, s, (uint32_t)TE_NAV_WAIT);
    return respawn;
  }
};

// cfg.drone_contact: armed drones as spheres of contact_radius, resolved once per env.step on the state the sub-step launch left, pairs
// in slot order, one pass (SEMANTICS.md; oracle: drone_contacts).  Compiled into the CONTACT instantiations of engage_kernel only: its 6 * DM
// registers would otherwise count against every launch (210 -> 363 VGPRs for <2, 9>) although the switch is off in every preset.
template <int DM>
TE_DEV void drone_contact_pass(const Params& p, int env, bool valid, uint64_t A) {
  const te_config& c = p.cfg;
  const int D = p.D;
  const EnvIO io(p, env);
  float qx[DM], qy[DM], qz[DM], ux[DM], uy[DM], uz[DM];
  uint64_t moved = 0u;
#pragma unroll
  for (int s = 0; s < DM; ++s) {
    qx[s] = qy[s] = qz[s] = ux[s] = uy[s] = uz[s] = 0.0f;
    if (s < D) {
      qx[s] = io.ldf(TE_D_POS, s); qy[s] = io.ldf(TE_D_POS + 1, s); qz[s] = io.ldf(TE_D_POS + 2, s);
      ux[s] = io.ldf(TE_D_VEL, s); uy[s] = io.ldf(TE_D_VEL + 1, s); uz[s] = io.ldf(TE_D_VEL + 2, s);
    }
  }
  const float two_r = 2.0f * c.contact_radius;
#pragma unroll
  for (int i = 0; i < DM; ++i) {
#pragma unroll
    for (int j = i + 1; j < DM; ++j) {
      float nx = qx[j] - qx[i], ny = qy[j] - qy[i], nz = qz[j] - qz[i];
      const float d = sqrtf(nx * nx + ny * ny + nz * nz);
      const bool hit = valid && ((A >> i) & (A >> j) & 1u) != 0u && d < two_r && d > 0.0f;
      const float inv = hit ? 1.0f / d : 0.0f;
      nx *= inv; ny *= inv; nz *= inv;
      const float push = hit ? 0.5f * (two_r - d) : 0.0f;
      const float vn = (ux[j] - ux[i]) * nx + (uy[j] - uy[i]) * ny + (uz[j] - uz[i]) * nz;
      const float dv = (hit && vn < 0.0f) ? 0.5f * vn : 0.0f;
      qx[i] -= push * nx; qy[i] -= push * ny; qz[i] -= push * nz; qx[j] += push * nx; qy[j] += push * ny; qz[j] += push * nz;
      ux[i] += dv * nx; uy[i] += dv * ny; uz[i] += dv * nz; ux[j] -= dv * nx; uy[j] -= dv * ny; uz[j] -= dv * nz;
      moved |= hit ? (((uint64_t)1 << i) | ((uint64_t)1 << j)) : (uint64_t)0;
    }
  }
#pragma unroll
  for (int s = 0; s < DM; ++s) {
    if ((moved >> s) & 1u) {
      io.stf(TE_D_POS, s, qx[s]); io.stf(TE_D_POS + 1, s, qy[s]); io.stf(TE_D_POS + 2, s, qz[s]);
      io.stf(TE_D_VEL, s, ux[s]); io.stf(TE_D_VEL + 1, s, uy[s]); io.stf(TE_D_VEL + 2, s, uz[s]);
    }
  }
}

template <int PM, int IM, bool CONTACT = false>
__global__ __launch_bounds__(64) void engage_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  TE_EXACT   // (see "exact arithmetic" above: the reward and the rows are computed by engage_slots_kernel too)
  constexpr int DM = PM + IM;
  using M = typename std::conditional<(DM > 32), uint64_t, uint32_t>::type;   // slot masks: 64 bits for Level5DumbMultiObs' 37 drones
  constexpr M one = 1;
  auto popc = [](M m) { return sizeof(M) == 8 ? __popcll((unsigned long long)m) : __popc((uint32_t)m); };
  auto lowest = [](M m) { return sizeof(M) == 8 ? __ffsll((long long)m) - 1 : __ffs((int)(uint32_t)m) - 1; };
  __shared__ float xs[DM][4];  // positions of the slots a wave has just respawned, handed back to the env's own lane
  __shared__ float rows[64 * TE_OBS_INERTIAL_WORDS];  // the chunk's [64, 15] inertial rows, transposed here so that they leave as 15 contiguous 256-byte stores
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers;
  const int lane = threadIdx.x;
  const int env = blockIdx.x * 64 + lane;
  const bool valid = env < p.N;   // planes are padded to Npad (a multiple of 64): lanes beyond N load in bounds and store nothing
  const GView g{p.dstate, p.estate, D, p.Npad, env, P};
  TE_ESTAMP(0, 0);
  const M pur_bits = (one << P) - one, all_bits = D >= (int)(8 * sizeof(M)) ? ~(M)0 : ((one << D) - one), inv_bits = all_bits & ~pur_bits;
  const bool scripted = all_scripted(c);

  // ---- one round of independent loads ----------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.dstate, 0, (int)((uint32_t)(TE_DRONE_WORDS + TE_X_WORDS) * (uint32_t)D * (uint32_t)p.Npad * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(p.estate, 0, (int)((uint32_t)TE_ENV_WORDS * (uint32_t)p.Npad * 4u), 0x00020000);
  const int voff = env * 4;
  const uint32_t plane = (uint32_t)p.Npad * 4u;   // bytes between two slots of a word; D * plane between two words
  auto ld = [&](int w, int s) { return __builtin_amdgcn_raw_buffer_load_b32(rd, voff, (int)(((uint32_t)w * (uint32_t)D + (uint32_t)s) * plane), 0); };
  auto le = [&](int w) { return __builtin_amdgcn_raw_buffer_load_b32(re, voff, (int)((uint32_t)w * plane), 0); };
  // stores: the same addressing (a 64-bit pointer per store costs ~5 VALU instructions of address arithmetic; this is one scalar
  // multiply).  st: word w of the wave-uniform slot s; stv: word w of a per-lane slot; ste: env word
  auto st = [&](int w, int s, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, rd, voff, (int)(((uint32_t)w * (uint32_t)D + (uint32_t)s) * plane), 0); };
  auto stf = [&](int w, int s, float v) { st(w, s, __float_as_uint(v)); };
  auto stv = [&](int w, int slot_off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, rd, slot_off, (int)((uint32_t)w * (uint32_t)D * plane), 0); };
  auto ste = [&](int w, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, re, voff, (int)((uint32_t)w * plane), 0); };
  auto stef = [&](int w, float v) { ste(w, __float_as_uint(v)); };
  float px[DM], py[DM], pz[DM];
  uint32_t armed_w[DM];
  int mun[PM], lf[PM];
  float fx[PM], fy[PM], fz[PM];   // FORMATION of the pursuers (the behaviour tree's MoveToFormation target)
  // live: slots armed in at least one env of this chunk, left with the flight plan by the previous launch (or the census after a reset /
  // te_set_state); the sub-step launch does not touch armed flags.  The four rows of an invader slot nobody has armed are not requested
  // (a disarmed drone's position is never looked at: S masks it) — 5-8 of stage03's 11 slots are live in a rollout: 20 -> ~10 MB fetched
  const uint32_t* __restrict__ lm32 = reinterpret_cast<const uint32_t*>(p.live_mask);
  const uint64_t live = (uint64_t)__builtin_amdgcn_readfirstlane(lm32[2 * blockIdx.x]) | ((uint64_t)__builtin_amdgcn_readfirstlane(lm32[2 * blockIdx.x + 1]) << 32);
#pragma unroll
  for (int s = 0; s < DM; ++s) {
    px[s] = py[s] = pz[s] = 0.0f; armed_w[s] = 0u;
    if (s < D && (s < P || ((live >> s) & 1u))) {
      px[s] = __uint_as_float(ld(TE_D_OBS_POS, s)); py[s] = __uint_as_float(ld(TE_D_OBS_POS + 1, s)); pz[s] = __uint_as_float(ld(TE_D_OBS_POS + 2, s));
      armed_w[s] = ld(TE_D_ARMED, s);
    }
  }
#pragma unroll
  for (int q = 0; q < PM; ++q) {
    mun[q] = 0; lf[q] = 0; fx[q] = fy[q] = fz[q] = 0.0f;
    if (q < P) {
      mun[q] = (int)ld(TE_D_MUNITION, q); lf[q] = (int)ld(TE_D_LAST_FIRED, q);
      fx[q] = __uint_as_float(ld(TE_D_FORMATION, q)); fy[q] = __uint_as_float(ld(TE_D_FORMATION + 1, q)); fz[q] = __uint_as_float(ld(TE_D_FORMATION + 2, q));
    }
  }
  float ag[9];  // OBS_EULER, OBS_VEL, OBS_RATE of the agent
#pragma unroll
  for (int k = 0; k < 9; ++k) ag[k] = __uint_as_float(ld(TE_D_OBS_EULER + k, 0));
  // level5: the wingmen's attitudes go into the snapshot planes below.  Requested HERE, with everything else: read where they are stored
  // (the planes may alias as far as the compiler knows) every one of the 3 * P words waited for the stores in front of it, 8 us of a 41 us launch
  uint32_t se[3][PM];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int q = 0; q < PM; ++q) se[k][q] = (p.snap && q < P) ? ld(TE_D_OBS_EULER + k, q) : 0u;
  int step = (int)le(TE_E_STEP) + 1;  // AGENT_STEP_BROADCAST (exp03_vFinal_environment.py:177-182)
  int max_step = (int)le(TE_E_MAX_STEP);
  int round = (int)le(TE_E_ROUND);
  const float last_dist = __uint_as_float(le(TE_E_LAST_DIST));
  int agent_kills = (int)le(TE_E_AGENT_KILLS), allies_kills = (int)le(TE_E_ALLIES_KILLS), deads = (int)le(TE_E_DEADS);
  uint32_t episode = le(TE_E_EPISODE);
  float4 act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (valid) act = reinterpret_cast<const float4*>(actions)[env];

  TE_ESTAMP(1, 1);
  // ---- masks, closest invader of every pursuer (OffsetHandler over the drones armed NOW, offsets_handler.py:68-95) -------
  // (branch-free on purpose: every pair is evaluated and masked, a few hundred VALU instructions; predicated skips cost more in
  // exec-mask bookkeeping than the arithmetic they save, and the lanes of a wave disagree about which slots are armed anyway)
  M S = 0, zone = 0, org = 0;
#pragma unroll
  for (int s = 0; s < DM; ++s) {
    const M a = (s < D && armed_w[s] != 0u && valid) ? one : (M)0;
    const float n = fnorm(V3{px[s], py[s], pz[s]});
    S |= a << s;
    zone |= (a & (n > c.dome_radius ? one : (M)0)) << s;
    org |= (a & (n < c.origin_range ? one : (M)0)) << s;
  }
  int tgt[PM]; float dmin[PM];
#pragma unroll
  for (int q = 0; q < PM; ++q) {
    tgt[q] = -1; dmin[q] = 0.0f;
#pragma unroll
    for (int j = 1; j < DM; ++j) {   // identify_closest_invader (offsets_handler.py:256-281): strict '<' in slot order
      const float d = fdist(V3{px[q], py[q], pz[q]}, V3{px[j], py[j], pz[j]});
      const bool take = q < P && j >= P && ((S >> q) & (S >> j) & one) != 0 && (tgt[q] < 0 || d < dmin[q]);
      tgt[q] = take ? j : tgt[q];
      dmin[q] = take ? d : dmin[q];
    }
  }
  auto pos_of = [&](int s) {  // position of a run-time slot: a select chain, never an indexed register file
    V3 r{0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < DM; ++k) if (k == s) r = V3{px[k], py[k], pz[k]};
    return r;
  };

  TE_ESTAMP(2, 0);
  M A = S;
  if (valid) {
    stef(TE_E_LAST_ACTION + 0, act.x); stef(TE_E_LAST_ACTION + 1, act.y); stef(TE_E_LAST_ACTION + 2, act.z); stef(TE_E_LAST_ACTION + 3, act.w);
    ste(TE_E_STEP, (uint32_t)step);
    ste(TE_E_SNAP_MASK, (uint32_t)S); ste(TE_E_SNAP_MASK_HI, (uint32_t)((uint64_t)S >> 32));
  }
  // a kill only clears the drone's bit here; Quadcopter.disarm's stores (quadcopter.py:461-478) are issued once per killed
  // drone after the engagement (ONE copy of the 15 stores in the code instead of one per call site), before anything respawns
  M killed = 0;
  auto kill = [&](int j) { killed |= one << j; A &= ~(one << j); };
  int agent_shots = 0, ally_shots = 0, exploded = 0, pursuer_suicided = 0, agent_suicided = 0;
  // process_shoot_range_invaders (exp03_vFinal_task.py:392-413)
#pragma unroll
  for (int q = 0; q < PM; ++q) {
    if (q < P && valid && ((S >> q) & one) && tgt[q] >= 0 && dmin[q] < c.shoot_range && gun_available(c, mun[q], lf[q], step) && mun[q] > 0) {
      mun[q] -= 1; lf[q] = step;
      st(TE_D_MUNITION, q, (uint32_t)mun[q]); st(TE_D_LAST_FIRED, q, (uint32_t)step);
      const U4 r = env_rng(c, env, RNG_HIT, (uint32_t)q, 0, episode, (uint32_t)step);
      if (u01(r.x) < c.hit_prob) {  // gun.py:94; entities_manager.shoot_by_ids (:238-248)
        kill(tgt[q]);
        if (q == 0) agent_shots += 1; else ally_shots += 1;
        if (c.evaluation) g.si(TE_D_KILLS, q, g.gi(TE_D_KILLS, q) + 1);  // lw_kills (evaluation_task.py:498-499)
      }
    }
  }
  // process_explosion_range_invaders (:359-390) on the same (stale) distances
#pragma unroll
  for (int q = 0; q < PM; ++q) {
    if (q < P && valid && ((S >> q) & one) && tgt[q] >= 0 && dmin[q] < c.explosion_range) {
      kill(q); kill(tgt[q]);
      if (mun[q] == 0 && q == 0) agent_suicided += 1;
      else if (mun[q] == 0) pursuer_suicided += 1;
      else exploded += 1;
    }
  }
  agent_kills += agent_shots; allies_kills += ally_shots; deads += exploded;
  // process_invaders_in_origin (:656-659); commented out in Evaluation_Task.on_step_middle (evaluation_task.py:397)
  if ((!c.evaluation || (c.evaluation & TE_EVAL_ORIGIN_RULE)) && valid) { killed |= org & inv_bits; A &= ~(org & inv_bits); }
  for (M m = killed; m; m &= m - 1) {   // Quadcopter.disarm: static body, velocities / motors / set-point zeroed
    const int so = (lowest(m) * p.Npad + env) * 4;
    stv(TE_D_ARMED, so, 0u);
#pragma unroll
    for (int k = 0; k < 3; ++k) { stv(TE_D_VEL + k, so, 0u); stv(TE_D_OMEGA + k, so, 0u); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { stv(TE_D_THROTTLE + k, so, 0u); stv(TE_D_SETPOINT + k, so, 0u); }
  }

  // ---- cfg.drone_contact (opt-in; stated model, parity with Bullet unpinned: include/threatengage.h)
  if (CONTACT) drone_contact_pass<DM>(p, env, valid, (uint64_t)A);
  const V3 apos{px[0], py[0], pz[0]};
  TE_ESTAMP(3, 0);
  // (the sphere's patches are the kernel's slowest stores — scattered 32-byte sectors, ~2.8 us per hit and env at 65 536 envs,
  // tools/patch_bench.hip — so they are issued as early as the data allows: right after the termination decision, which picks their
  // buffer; the reward is worked out while they drain)
  // increment_max_step (:150-153), compute_termination (:517-569)
  if (agent_shots + ally_shots > 0) max_step += c.step_increment;
  const int armed_invaders = popc(A & inv_bits), armed_pursuers = popc(A & pur_bits);
  const bool all_rounds_over = armed_invaders == 0 && round >= c.n_rounds;
  bool term;
  if (c.evaluation) term = (c.max_step > 0 && step > max_step) || all_rounds_over || zone != 0u || armed_pursuers == 0;
  else term = step > max_step || all_rounds_over || zone != 0u || armed_pursuers == 0 || (c.agent_death_terminates && !(A & one)) || apos.z < -5.99f;
  const bool to_terminal = valid && term && c.auto_reset;
  TE_ESTAMP(4, 0);
  // ---- the agent's own sphere: LidarMath.reframe + binning of every other drone armed NOW, closer wins in slot order
  // (fused_lidar.py:143-217, lidar_math.py:53-83,262-311); empty right after a reset (step 0 never gets here)
  // (a stacked-observation env has no own-sphere output at all — ring_push_kernel / stack_view_kernel draw its spheres from the snapshot
  // planes below — and skips the binning on a scalar test: 17 to 36 asin / atan2 pairs per lane that nothing reads)
  M owners = 0;
  uint32_t cell[DM]; float rhat[DM];
#pragma unroll
  for (int j = 0; j < DM; ++j) { cell[j] = 0u; rhat[j] = 1.0f; }
  if (o.obs.lidar || o.term.lidar) {
    const M3 R = x_inverse_attitude(ag[0], ag[1], ag[2]);
#pragma unroll
    for (int j = 1; j < DM; ++j) {   // a slot somebody of the chunk has armed is binned in every lane (branch-free, as above); only armed ones may own a cell
      if (!(j < D && ((live >> j) & 1u))) continue;   // wave-uniform: nobody has it armed, nobody looks at its cell (owners stays clear)
      int cj; lidar_cell_fast(c, x_mul(R, sub(V3{px[j], py[j], pz[j]}, apos)), cj, rhat[j]);
      cell[j] = (uint32_t)cj;
      const bool in = ((A >> j) & one) != 0;
      M same = 0;           // the current owner of the same cell, if any (at most one)
      float r_owner = 2.0f;
#pragma unroll
      for (int k = 1; k < j; ++k) {   // (a pair costs four VALU instructions; a compare + scalar branch per pair, with the bookkeeping only for
                                      // pairs that do collide in some lane, was SLOWER here: + 3 us at DM = 11 and 32, a lone wave pays for every branch)
        const bool hit = ((owners >> k) & one) != 0 && cell[k] == cell[j];
        same |= (hit ? one : (M)0) << k;
        r_owner = hit ? rhat[k] : r_owner;
      }
      const bool wins = in && (same ? rhat[j] < r_owner : rhat[j] < 1.0f);   // an empty cell holds 1.0
      owners = wins ? ((owners & ~same) | (one << j)) : owners;
    }
  }
  TE_ESTAMP(5, 0);
  // ---- level5: what this step's stacked observation may look at, BEFORE anything respawns (te_stacked.hpp SnapRows)
  if (p.snap && valid) {
    const SnapRows sr{D, P};
#pragma unroll
    for (int s = 0; s < DM; ++s) {
      if (s < D) {
        p.snap[(size_t)(sr.pos() + 0 * D + s) * p.Npad + env] = __float_as_uint(px[s]);
        p.snap[(size_t)(sr.pos() + 1 * D + s) * p.Npad + env] = __float_as_uint(py[s]);
        p.snap[(size_t)(sr.pos() + 2 * D + s) * p.Npad + env] = __float_as_uint(pz[s]);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int s = 0; s < PM; ++s)
        if (s < P) p.snap[(size_t)(sr.euler() + k * P + s) * p.Npad + env] = se[k][s];
    p.snap[(size_t)sr.armed() * p.Npad + env] = (uint32_t)A; p.snap[(size_t)sr.armed_hi() * p.Npad + env] = (uint32_t)((uint64_t)A >> 32);
    p.snap[(size_t)sr.step() * p.Npad + env] = (uint32_t)step;
    p.snap[(size_t)sr.episode() * p.Npad + env] = episode;
    p.snap[(size_t)sr.done() * p.Npad + env] = to_terminal ? 1u : 0u;
  }
  TE_ESTAMP(12, 0);
  // ---- terminal observation of an auto-reset env (SB3 VecEnv: infos[i]["terminal_observation"]): rows from this lane
  auto inertial_row = [&](float* dst, float x, float y, float z, const float a9[9], int mu, int lfi, int st) {
    const float two_pi = 2.0f * kPi;
    dst[0] = clampf(x / c.dome_radius, -1.0f, 1.0f); dst[1] = clampf(y / c.dome_radius, -1.0f, 1.0f); dst[2] = clampf(z / c.dome_radius, -1.0f, 1.0f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dst[3 + k] = clampf(a9[3 + k] / c.max_speed, -1.0f, 1.0f);
      dst[6 + k] = clampf(a9[k] / kPi, -1.0f, 1.0f);
      dst[9 + k] = clampf(a9[6 + k] / two_pi, -1.0f, 1.0f);
    }
    float gs[3];
    gun_state(c, mu, lfi, st, max_munition_of(c, 0), gs);
    dst[12] = gs[0]; dst[13] = gs[1]; dst[14] = gs[2];
  };
  if (to_terminal) {
    if (o.term.inertial) inertial_row(o.term.inertial + (size_t)env * TE_OBS_INERTIAL_WORDS, px[0], py[0], pz[0], ag, mun[0], lf[0], step);
    if (o.term.last_action) reinterpret_cast<float4*>(o.term.last_action)[env] = act;
  }
  // terminal LIDAR tiles: ones by the whole wave, env by env (rare).  The patches below come from the SAME wave, later in program
  // order: the stores of one wave to one address are performed in issue order, so nothing has to be drained in between
  if (o.term.lidar) {
    for (unsigned long long tb = __ballot(to_terminal); tb; tb &= tb - 1) {
      const int l = __ffsll((long long)tb) - 1;
      float* tile = o.term.lidar + (size_t)(blockIdx.x * 64 + l) * lidar_words(c);
      for (int e = lane; e < lidar_words(c); e += 64) tile[e] = 1.0f;
    }
  }
  // patch the hit cells into the background the sub-step kernel's fill waves wrote (lidar_math.py:305: flag = type / 5;
  // time plane = Delta 1 of a 10-deep ring, perception_snapshot.py:36-37)
  {
    float* dst = to_terminal ? o.term.lidar : o.obs.lidar;
    if (valid && dst) {
      dst += (size_t)env * lidar_words(c);
      const bool time_plane = c.lidar_channels != 2;
      // plane by plane, not hit by hit: the hits of one plane fall into the same 1.3 KB of the env's tile, and consecutive store
      // instructions that stay within it drain faster (interleaved A/B, round 3: engage kernel 29.4 -> 28.7 us, 49.8 -> 47 us with every slot armed)
#pragma unroll
      for (int j = 1; j < DM; ++j) if ((owners >> j) & one) dst[cell[j]] = rhat[j];
#pragma unroll
      for (int j = 1; j < DM; ++j) if ((owners >> j) & one) dst[TE_LIDAR_CELLS + cell[j]] = (float)(j < P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;
      if (time_plane) {
#pragma unroll
        for (int j = 1; j < DM; ++j) if ((owners >> j) & one) dst[2 * TE_LIDAR_CELLS + cell[j]] = 0.1f;
      }
    }
    if (valid && o.persist && o.obs.lidar) {   // persistent observation: which cells of the MAIN buffer hold a feature now (an auto-reset env shows the empty sphere)
      uint16_t* pv = o.prev + env;
      int n = 0;
#pragma unroll
      for (int j = 1; j < DM; ++j)
        if (!to_terminal && ((owners >> j) & one)) { n += 1; pv[(size_t)n * p.Npad] = (uint16_t)cell[j]; }
      pv[0] = (uint16_t)n;
    }
  }

  TE_ESTAMP(13, 0);
  // compute_reward (:423-515); Evaluation_Task.compute_reward returns 0 (evaluation_task.py:508-515)
  float reward = 0.0f, cur_dist = last_dist;
  if (!c.evaluation) {
    float gs[3];
    gun_state(c, mun[0], lf[0], step, max_munition_of(c, 0), gs);
    const float dist_origin = fnorm(apos);
    int ally = -1;  // identify_closest_ally (offsets_handler.py:167-190)
    if ((S & one) && popc(S & pur_bits) > 1) {
      float bd = 0.0f;
#pragma unroll
      for (int a = 1; a < PM; ++a) {
        if (a < P && ((S >> a) & one)) {
          const float d = fdist(V3{px[a], py[a], pz[a]}, apos);
          if (ally < 0 || d < bd) { ally = a; bd = d; }
        }
      }
    }
    int target = -1;
#pragma unroll
    for (int q = 0; q < PM; ++q) if (q == (ally < 0 ? 0 : ally) && ((S >> q) & one)) target = tgt[q];
    const V3 tp = target >= 0 ? pos_of(target) : V3{0.0f, 0.0f, 0.0f};
    cur_dist = fdist(apos, tp);
    const bool ready = gs[2] == 1.0f || gs[0] == 0.0f;
    float bonus = 0.0f, penalty = 0.0f;
    if (c.reward_model == TE_REWARD_L5_DUMB) {  // Level5DumbMultiObjectTask.compute_reward (level5_dumb_multiobject_task.py:452-553)
      const float SAFE = 5.0f;
      float score = -cur_dist;
      if (!ready) {                               // keep away while reloading: distance counts for, closeness against
        score = cur_dist;
        if (cur_dist < SAFE) penalty += (SAFE - cur_dist) / SAFE * 500.0f;
      }
      if (gs[2] == 0.0f && gs[0] > 0.0f && (cur_dist - last_dist) > 0.01f) bonus += 100.0f;
      if (agent_shots > 0) bonus += (float)agent_shots * 1000.0f;
      if (ally_shots > 0 || pursuer_suicided > 0) bonus += 0.5f * (float)(ally_shots + pursuer_suicided) * 1000.0f;
      if (agent_suicided > 0) penalty += 2.0f * (float)agent_suicided * 1000.0f;
      if (exploded > 0) penalty += 1000.0f * (float)exploded;
      if (apos.z < -5.0f) penalty += fminf(-5.0f - apos.z, 1.0f) * 1000.0f;
      if (zone & pur_bits) penalty += 1000.0f;
      if (dist_origin > c.born_radius - 2.0f) penalty += fminf(dist_origin - (c.born_radius - 2.0f), 1000.0f);
      reward = clampf(score + bonus - penalty, -3000.0f, 3000.0f);
    } else if (c.reward_model == TE_REWARD_L5_C1) {  // Level5C1FusionTask.compute_reward (level5_c1_fusion_task.py:448-485)
      const int t1 = (S & one) ? tgt[0] : -1;        // the agent's OWN closest invader (:458)
      const float d1 = t1 >= 0 ? fdist(apos, pos_of(t1)) : dist_origin;
      const float first = last_dist == 0.0f ? d1 : last_dist;   // `last_distance` is set by the first reward of the env and never again (:467-468)
      float r1 = d1 < first ? c.approach_bonus_gain * fnorm(V3{ag[3], ag[4], ag[5]}) : 0.0f;
      if (agent_shots > 0) r1 += (float)agent_shots * 1000.0f;
      if (agent_suicided > 0) r1 -= 2.0f * (float)agent_suicided * 1000.0f;
      reward = clampf(r1, -3000.0f, 3000.0f);
      cur_dist = first;
    } else {
    if (0.01f < last_dist - cur_dist && ready) bonus += c.approach_bonus_gain * fnorm(V3{ag[3], ag[4], ag[5]});
    const float score = ready ? -cur_dist : cur_dist * (2.0f * gs[1] - 1.0f);
    if (agent_shots > 0 || agent_suicided > 0) bonus += (float)(agent_shots + agent_suicided) * 1000.0f;
    if (ally_shots > 0 || pursuer_suicided > 0) bonus += 0.5f * (float)(ally_shots + pursuer_suicided) * 1000.0f;
    else if (exploded > 0) penalty += 1000.0f * (float)exploded;
    if (apos.z < -5.0f) penalty += (-5.0f - apos.z) * 1000.0f;
    if (zone & pur_bits) penalty += 1000.0f;
    if (dist_origin > c.born_radius - 2.0f) penalty += dist_origin - c.born_radius - 2.0f;  // literal
    reward = score + bonus - penalty;
    }
  }
  TE_ESTAMP(14, 0);
  if (valid) {
    if (!c.evaluation) stef(TE_E_LAST_DIST, cur_dist);
    ste(TE_E_AGENT_KILLS, (uint32_t)agent_kills); ste(TE_E_ALLIES_KILLS, (uint32_t)allies_kills); ste(TE_E_DEADS, (uint32_t)deads);
    if (agent_shots + ally_shots > 0) ste(TE_E_MAX_STEP, (uint32_t)max_step);
    o.reward[env] = reward;
    o.done[env] = term ? 1 : 0;
    reinterpret_cast<int4*>(o.info)[env] = make_int4(agent_kills, allies_kills, deads, round);
    ste(TE_E_INFO_WAVE, (uint32_t)round);   // (on_step_end below may start the next wave; an auto-reset puts 1 back)
  }

  TE_ESTAMP(6, 0);
  // ---- on_step_end (:321-333): next wave when this one is cleared and a pursuer is alive; SB3 auto-reset -------------------
  uint32_t task = 0u;   // round | reset << 8: the slots of this env have to be respawned
  M snap_mask = S;
  auto mask_after_spawn = [&](int rnd, bool reset) {
    const M m = reset ? pur_bits : (A & pur_bits);
    const int n = invaders_in_round(c, rnd);
    return (M)(m | ((((one << n) - one) << P) & all_bits));
  };
  if (valid && !term && armed_invaders == 0 && armed_pursuers > 0) {
    round = round + (round < c.n_rounds ? 1 : c.n_rounds);  // advance_round (:155-175)
    snap_mask = mask_after_spawn(round, false);
    ste(TE_E_ROUND, (uint32_t)round); ste(TE_E_SNAP_MASK, (uint32_t)snap_mask); ste(TE_E_SNAP_MASK_HI, (uint32_t)((uint64_t)snap_mask >> 32));
    task = (uint32_t)round;
  }
  if (to_terminal) {  // Env.reset -> Task.on_reset (exp03_vFinal_environment.py:128-146): the env record here, the slots below
    episode += 1u; step = 0; max_step = c.max_step; round = 1;
    snap_mask = mask_after_spawn(1, true);
    ste(TE_E_EPISODE, episode); ste(TE_E_STEP, 0u); ste(TE_E_MAX_STEP, (uint32_t)c.max_step); ste(TE_E_ROUND, 1u); ste(TE_E_INFO_WAVE, 1u);
    ste(TE_E_AGENT_KILLS, 0u); ste(TE_E_ALLIES_KILLS, 0u); ste(TE_E_DEADS, 0u);
    if (c.reward_model != TE_REWARD_L5_C1) stef(TE_E_LAST_DIST, c.dome_radius);
#pragma unroll
    for (int k = 0; k < 4; ++k) ste(TE_E_LAST_ACTION + k, 0u);
    ste(TE_E_SNAP_MASK, (uint32_t)snap_mask); ste(TE_E_SNAP_MASK_HI, (uint32_t)((uint64_t)snap_mask >> 32));
    act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) ag[k] = 0.0f;
    task = 1u | (1u << 8);
  }
  // ---- spawn: the D slots of a respawning env are taken by the lanes of the wave (lane k = slot k), env by env -------------
  M armed_post = A;
  {
    unsigned long long sb = __ballot(task != 0u);
    for (; sb; sb &= sb - 1) {
      const int l = __ffsll((long long)sb) - 1;
      const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)task, l);
      const uint32_t ep = (uint32_t)__builtin_amdgcn_readlane((int)episode, l);
      const M armed_l = (M)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)A, l) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)A >> 32), l) << 32));   // the env's flags after the engagement
      const bool reset = (t >> 8) != 0u;
      if (lane < D) {
        const GView gv{p.dstate, p.estate, D, p.Npad, (int)(blockIdx.x * 64 + l), P};
        V3 w{0.0f, 0.0f, 0.0f};
        const bool placed = spawn_slot_at(c, gv, lane, (int)(t & 0xFFu), ep, reset, ((armed_l >> lane) & one) != 0, w);
        xs[lane][0] = w.x; xs[lane][1] = w.y; xs[lane][2] = w.z; xs[lane][3] = placed ? 1.0f : 0.0f;
      }
      // single-wave workgroup: the LDS operations of one wave execute in order; only the compiler and the LDS counter have to be
      // told (a __syncthreads() would also drain every outstanding global store of the wave: several microseconds here)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == l) {
#pragma unroll
        for (int s = 0; s < DM; ++s)
          if (s < D && xs[s][3] != 0.0f) { px[s] = xs[s][0]; py[s] = xs[s][1]; pz[s] = xs[s][2]; }
        armed_post = snap_mask;   // = the mask after the spawn: the pursuers as they are (all armed on reset) + the round's invaders
        if (reset) {
#pragma unroll
          for (int q = 0; q < PM; ++q)
            if (q < P) { mun[q] = max_munition_of(c, q); lf[q] = -c.cooldown_steps; fx[q] = px[q]; fy[q] = py[q]; fz[q] = pz[q]; }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  TE_ESTAMP(7, 0);
  // ---- the observation of the state the step leaves (post-reset values for an auto-reset env)
  if (o.obs.inertial) {
    float row15[TE_OBS_INERTIAL_WORDS];
    inertial_row(row15, px[0], py[0], pz[0], ag, mun[0], lf[0], step);
#pragma unroll
    for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) rows[lane * TE_OBS_INERTIAL_WORDS + k] = row15[k];   // stride 15: conflict-free
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int n_out = min(64, p.N - (int)blockIdx.x * 64) * TE_OBS_INERTIAL_WORDS;
    float* dst = o.obs.inertial + (size_t)blockIdx.x * 64 * TE_OBS_INERTIAL_WORDS;
#pragma unroll
    for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k)
      if (k * 64 + lane < n_out) dst[k * 64 + lane] = rows[k * 64 + lane];
  }
  if (valid && o.obs.last_action) reinterpret_cast<float4*>(o.obs.last_action)[env] = act;
  TE_ESTAMP(8, 0);
  // ---- the pursuers' positions the invaders steer at during the next sub-step launch (TE_X_REF), and the scripted allies'
  // commands of the next step (Task.on_step_start -> LoyalWingmanBehaviorTree.update, loyalwingman_navigator.py:238-352)
  if (valid) {
    int first_skipped = -1;  // drive_loyalwingmen: get_armed_pursuers()[1:] — with the agent dead the first armed ally is skipped
    if (!scripted && !(armed_post & one)) first_skipped = (armed_post & pur_bits & ~one) ? lowest(armed_post & pur_bits & ~one) : -1;
#pragma unroll
    for (int q = 0; q < PM; ++q) {
      if (q < P) {
        stf(TE_X_REF + 0, q, px[q]); stf(TE_X_REF + 1, q, py[q]); stf(TE_X_REF + 2, q, pz[q]);
        if ((q > 0 || scripted) && ((armed_post >> q) & one) && q != first_skipped) {
          float out[3] = {0.0f, 0.0f, 0.0f};
          const bool ext = driven_externally(c, q);
          if (!ext && c.ally_policy == TE_ALLY_BT) {
            const V3 me{px[q], py[q], pz[q]};
            if (gun_available(c, mun[q], lf[q], step)) {
              int t = -1; float bd = 0.0f; V3 tp{0.0f, 0.0f, 0.0f};
              if ((snap_mask >> q) & one) {
#pragma unroll
                for (int j = 1; j < DM; ++j) {
                  if (j >= P && ((snap_mask >> j) & one)) {
                    const V3 pj{px[j], py[j], pz[j]};
                    const float d = fdist(me, pj);
                    if (t < 0 || d < bd) { t = j; bd = d; tp = pj; }
                  }
                }
              }
              x_cmd_toward(me, tp, c.ally_speed, out);
            } else x_cmd_toward(me, V3{fx[q], fy[q], fz[q]}, c.ally_speed, out);
          } else if (ext || c.ally_policy != TE_ALLY_FROZEN) {  // nobody / the caller's policy: the set-point persists
            out[0] = g.gf(TE_D_SETPOINT + 0, q); out[1] = g.gf(TE_D_SETPOINT + 1, q); out[2] = g.gf(TE_D_SETPOINT + 3, q);
          }
          stf(TE_X_CMD + 0, q, out[0]); stf(TE_X_CMD + 1, q, out[1]); stf(TE_X_CMD + 2, q, out[2]);
        }
      }
    }
  }
  TE_ESTAMP(9, 0);
  // ---- what the next sub-step launch has to fly for this chunk (post-spawn flags)
  {
    uint64_t dense = 0u, live = 0u; int n = 0;
    uint16_t* items = p.mixed_items + (size_t)blockIdx.x * kMixedCap;
    for (int s = 0; s < D; ++s) plan_slot_l4<void>(items, p.dense_min, lane, s, valid && ((armed_post >> s) & one) != 0, dense, n, live);
    if (lane == 0) { p.slot_mask[blockIdx.x] = dense; p.mixed_count[blockIdx.x] = (uint32_t)n; p.live_mask[blockIdx.x] = live; }
  }
  TE_ESTAMP(10, 0);
  TE_ESTAMP(11, 1);   // ... and once every store has been acknowledged
}

}  // namespace te

// =================================================================================================================================
// stage01 / stage02 in the same form (one lane per env, one wave per chunk, the env in registers): BASELINE configs 2 and 3 run a
// few thousand envs per GPU, where the LDS kernel's phase chain (18 / 42 us at 4 096 / 16 384 envs) is pure latency.
// =================================================================================================================================
namespace te {

// the agent's own sphere over the drones armed in A (closer wins in slot order), from register positions: owners + cells + ranges
template <int DM>
TE_DEV uint32_t own_sphere_regs(const te_config& c, int D, const float (&px)[DM], const float (&py)[DM], const float (&pz)[DM], const float ag[9],
                                uint32_t A, uint32_t (&cell)[DM], float (&rhat)[DM]) {
  uint32_t owners = 0u;
  const V3 apos{px[0], py[0], pz[0]};
  const M3 R = x_inverse_attitude(ag[0], ag[1], ag[2]);
  cell[0] = 0u; rhat[0] = 1.0f;
#pragma unroll
  for (int j = 1; j < DM; ++j) {
    int cj; lidar_cell_fast(c, x_mul(R, sub(V3{px[j], py[j], pz[j]}, apos)), cj, rhat[j]);
    cell[j] = (uint32_t)cj;
    const bool in = j < D && ((A >> j) & 1u) != 0u;
    uint32_t same = 0u; float r_owner = 2.0f;
#pragma unroll
    for (int k = 1; k < j; ++k) {
      const bool hit = ((owners >> k) & 1u) != 0u && cell[k] == cell[j];
      same |= (hit ? 1u : 0u) << k;
      r_owner = hit ? rhat[k] : r_owner;
    }
    const bool wins = in && (same ? rhat[j] < r_owner : rhat[j] < 1.0f);
    owners = wins ? ((owners & ~same) | (1u << j)) : owners;
  }
  return owners;
}
// persistent observation: the cells of the main buffer that hold a feature after this step (see StepOut)
template <int DM>
TE_DEV void record_sphere_regs(const StepOut& o, int npad, int env, bool to_terminal, uint32_t owners, const uint32_t (&cell)[DM]) {
  if (!o.persist || !o.obs.lidar) return;
  uint16_t* pv = o.prev + env;
  int n = 0;
#pragma unroll
  for (int j = 1; j < DM; ++j)
    if (!to_terminal && ((owners >> j) & 1u)) { n += 1; pv[(size_t)n * npad] = (uint16_t)cell[j]; }
  pv[0] = (uint16_t)n;
}
template <int DM>
TE_DEV void patch_sphere_regs(const te_config& c, float* __restrict__ dst, int env, int P, uint32_t owners, const uint32_t (&cell)[DM], const float (&rhat)[DM]) {
  if (!dst) return;
  dst += (size_t)env * lidar_words(c);
  const bool time_plane = c.lidar_channels != 2;
#pragma unroll
  for (int j = 1; j < DM; ++j) if ((owners >> j) & 1u) dst[cell[j]] = rhat[j];   // plane by plane (see engage_kernel)
#pragma unroll
  for (int j = 1; j < DM; ++j) if ((owners >> j) & 1u) dst[TE_LIDAR_CELLS + cell[j]] = (float)(j < P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;
  if (time_plane) {
#pragma unroll
    for (int j = 1; j < DM; ++j) if ((owners >> j) & 1u) dst[2 * TE_LIDAR_CELLS + cell[j]] = 0.1f;
  }
}
// normalize_inertial_data + gun state of the agent (level4/components/utils/normalization.py:6-30,61-110; gun.py:101-113)
TE_DEV void inertial_row_regs(const te_config& c, float* dst, float x, float y, float z, const float a9[9], int mu, int lfi, int st) {
  const float two_pi = 2.0f * kPi;
  dst[0] = clampf(x / c.dome_radius, -1.0f, 1.0f); dst[1] = clampf(y / c.dome_radius, -1.0f, 1.0f); dst[2] = clampf(z / c.dome_radius, -1.0f, 1.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dst[3 + k] = clampf(a9[3 + k] / c.max_speed, -1.0f, 1.0f);
    dst[6 + k] = clampf(a9[k] / kPi, -1.0f, 1.0f);
    dst[9 + k] = clampf(a9[6 + k] / two_pi, -1.0f, 1.0f);
  }
  float gs[3];
  gun_state(c, mu, lfi, st, max_munition_of(c, 0), gs);
  dst[12] = gs[0]; dst[13] = gs[1]; dst[14] = gs[2];
}
// the chunk's [64, 15] rows through LDS: 15 contiguous 256-byte stores; last_action straight from the lane
TE_DEV void write_rows_regs(const Params& p, const ObsOut& o, float (&rows)[64 * TE_OBS_INERTIAL_WORDS], int lane, bool valid, int env, const float row[TE_OBS_INERTIAL_WORDS], float4 act) {
  if (o.inertial) {
#pragma unroll
    for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) rows[lane * TE_OBS_INERTIAL_WORDS + k] = row[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int n_out = min(64, p.N - (int)blockIdx.x * 64) * TE_OBS_INERTIAL_WORDS;
    float* dst = o.inertial + (size_t)blockIdx.x * 64 * TE_OBS_INERTIAL_WORDS;
#pragma unroll
    for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k)
      if (k * 64 + lane < n_out) dst[k * 64 + lane] = rows[k * 64 + lane];
  }
  if (valid && o.last_action) reinterpret_cast<float4*>(o.last_action)[env] = act;
}
TE_DEV void terminal_tiles(const te_config& c, float* t_lidar, bool to_terminal, int lane) {
  if (!t_lidar) return;
  for (unsigned long long tb = __ballot(to_terminal); tb; tb &= tb - 1) {
    const int l = __ffsll((long long)tb) - 1;
    float* tile = t_lidar + (size_t)(blockIdx.x * 64 + l) * lidar_words(c);
    for (int e = lane; e < lidar_words(c); e += 64) tile[e] = 1.0f;
  }
}

// ---- stage02: L3Stage1.on_step_middle / on_step_end (level3/components/stages.py:144-179,241-344), the restatement of stage02_logic
template <int PM, int IM>
__global__ __launch_bounds__(64) void engage_stage02_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  TE_EXACT   // (the reward is computed by engage_slots_stage02_kernel too: te_device.hpp "exact arithmetic")
  constexpr int DM = PM + IM;
  __shared__ float rows[64 * TE_OBS_INERTIAL_WORDS];
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers;
  const int lane = threadIdx.x, env = blockIdx.x * 64 + lane;
  const bool valid = env < p.N;
  const EnvIO io(p, env);
  const uint32_t pur_bits = (1u << P) - 1u, all_bits = D >= 32 ? 0xFFFFFFFFu : ((1u << D) - 1u), inv_bits = all_bits & ~pur_bits;
  float px[DM], py[DM], pz[DM]; uint32_t armed_w[DM];
  int mun[PM], lf[PM];
#pragma unroll
  for (int s = 0; s < DM; ++s) {
    px[s] = py[s] = pz[s] = 0.0f; armed_w[s] = 0u;
    if (s < D) { px[s] = io.ldf(TE_D_OBS_POS, s); py[s] = io.ldf(TE_D_OBS_POS + 1, s); pz[s] = io.ldf(TE_D_OBS_POS + 2, s); armed_w[s] = io.ld(TE_D_ARMED, s); }
  }
#pragma unroll
  for (int q = 0; q < PM; ++q) { mun[q] = 0; lf[q] = 0; if (q < P) { mun[q] = (int)io.ld(TE_D_MUNITION, q); lf[q] = (int)io.ld(TE_D_LAST_FIRED, q); } }
  float ag[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
  int step = (int)io.le(TE_E_STEP) + 1;
  const int max_step = (int)io.le(TE_E_MAX_STEP);
  int kills = (int)io.le(TE_E_AGENT_KILLS), deads = (int)io.le(TE_E_DEADS);
  uint32_t episode = io.le(TE_E_EPISODE);
  const float last = __uint_as_float(io.le(TE_E_PREV_SNAP_MIN));
  float4 act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (valid) act = reinterpret_cast<const float4*>(actions)[env];

  uint32_t S = 0u, zone = 0u;
#pragma unroll
  for (int s = 0; s < DM; ++s) {
    const uint32_t a = (s < D && armed_w[s] != 0u && valid) ? 1u : 0u;
    S |= a << s;
    zone |= (a & (fnorm(V3{px[s], py[s], pz[s]}) > c.dome_radius ? 1u : 0u)) << s;
  }
  int tgt[PM]; float dmin[PM];
#pragma unroll
  for (int q = 0; q < PM; ++q) {
    tgt[q] = -1; dmin[q] = 0.0f;
#pragma unroll
    for (int j = 1; j < DM; ++j) {
      const float d = fdist(V3{px[q], py[q], pz[q]}, V3{px[j], py[j], pz[j]});
      const bool take = q < P && j >= P && ((S >> q) & (S >> j) & 1u) != 0u && (tgt[q] < 0 || d < dmin[q]);
      tgt[q] = take ? j : tgt[q]; dmin[q] = take ? d : dmin[q];
    }
  }
  // stage02_agent_min_distance: the first armed pursuer's closest invader (0 when there is none), before any respawn
  float cur = 0.0f;
  {
    bool found = false;
#pragma unroll
    for (int q = 0; q < PM; ++q) if (!found && q < P && ((S >> q) & 1u)) { found = true; cur = tgt[q] >= 0 ? dmin[q] : 0.0f; }
  }
  uint32_t A = S, killed = 0u;
  if (valid) {
    io.stef(TE_E_LAST_ACTION + 0, act.x); io.stef(TE_E_LAST_ACTION + 1, act.y); io.stef(TE_E_LAST_ACTION + 2, act.z); io.stef(TE_E_LAST_ACTION + 3, act.w);
    io.ste(TE_E_STEP, (uint32_t)step); io.ste(TE_E_SNAP_MASK, S);
  }
  int shots = 0, exploded = 0;
#pragma unroll
  for (int q = 0; q < PM; ++q) {   // shoot_by_ids with the suicide rule (level3/components/quadcopter_manager.py:155-171)
    if (q < P && valid && ((S >> q) & 1u) && tgt[q] >= 0 && dmin[q] < c.shoot_range) {
      if (mun[q] == 0) { killed |= 1u << tgt[q]; shots += 1; }
      else if (gun_available(c, mun[q], lf[q], step)) {
        mun[q] -= 1; lf[q] = step;
        io.st(TE_D_MUNITION, q, (uint32_t)mun[q]); io.st(TE_D_LAST_FIRED, q, (uint32_t)step);
        const U4 r = env_rng(c, env, RNG_HIT, (uint32_t)q, 0, episode, (uint32_t)step);
        if (u01(r.x) < c.hit_prob) { killed |= 1u << tgt[q]; shots += 1; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PM; ++q)
    if (q < P && valid && ((S >> q) & 1u) && tgt[q] >= 0 && dmin[q] < c.explosion_range) { killed |= (1u << q) | (1u << tgt[q]); exploded += 1; }
  A &= ~killed;
  for (uint32_t m = killed; m; m &= m - 1) io.disarm(__ffs(m) - 1);
  kills += shots; deads += exploded;
  float gs[3];
  gun_state(c, mun[0], lf[0], step, max_munition_of(c, 0), gs);
  const bool ready = gs[2] == 1.0f || gs[0] == 0.0f;
  float bonus = 0.0f, penalty = 0.0f;
  const float score = ready ? -cur : cur * (2.0f * gs[1] - 1.0f);
  if (0.01f < last - cur && ready) bonus += c.approach_bonus_gain * fnorm(V3{ag[3], ag[4], ag[5]});
  bonus += 1000.0f * (float)shots; penalty += 1000.0f * (float)exploded;
  const bool outside_p = (zone & pur_bits) != 0u, outside_i = (zone & inv_bits) != 0u;
  if (outside_p) penalty += 1000.0f;
  const float reward = score + bonus - penalty;
  const bool term = step > max_step || outside_p || outside_i || __popc(A & pur_bits) < P;
  const bool to_terminal = valid && term && c.auto_reset;
  if (valid) {
    io.ste(TE_E_AGENT_KILLS, (uint32_t)kills); io.ste(TE_E_DEADS, (uint32_t)deads);
    o.reward[env] = reward; o.done[env] = term ? 1 : 0;
    reinterpret_cast<int4*>(o.info)[env] = make_int4(kills, 0, deads, 0);
  }
  // the observation is taken before the respawn: a drone armed after the step broadcast has no Delta = 1 snapshot yet
  uint32_t cell[DM]; float rhat[DM];
  const uint32_t owners = own_sphere_regs<DM>(c, D, px, py, pz, ag, A, cell, rhat);
  float row[TE_OBS_INERTIAL_WORDS];
  if (to_terminal) {
    inertial_row_regs(c, row, px[0], py[0], pz[0], ag, mun[0], lf[0], step);
    if (o.term.inertial) { for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) o.term.inertial[(size_t)env * TE_OBS_INERTIAL_WORDS + k] = row[k]; }
    if (o.term.last_action) reinterpret_cast<float4*>(o.term.last_action)[env] = act;
  }
  terminal_tiles(c, o.term.lidar, to_terminal, lane);
  if (valid) { patch_sphere_regs<DM>(c, to_terminal ? o.term.lidar : o.obs.lidar, env, P, owners, cell, rhat); record_sphere_regs<DM>(o, p.Npad, env, to_terminal, owners, cell); }
  // on_step_end: killed invaders come back (stages.py:167-174); last_offsets = current_offsets
  auto set_pos = [&](int s, V3 w) {
#pragma unroll
    for (int k = 0; k < DM; ++k) if (k == s) { px[k] = w.x; py[k] = w.y; pz[k] = w.z; }
  };
  if (valid) {
    for (uint32_t m = inv_bits & ~A; m; m &= m - 1) {
      const int j = __ffs(m) - 1;
      const V3 w = stage02_invader_position(c, env, j, episode, (uint32_t)step);
      io.respawn(c, j, w); set_pos(j, w);
    }
    A |= inv_bits;
    io.stef(TE_E_PREV_SNAP_MIN, cur); io.stef(TE_E_LAST_DIST, cur);
  }
  if (to_terminal) {  // L3Stage1.on_reset (stages.py:104-131)
    episode += 1u; step = 0;
    io.ste(TE_E_EPISODE, episode); io.ste(TE_E_STEP, 0u); io.ste(TE_E_MAX_STEP, (uint32_t)c.max_step); io.ste(TE_E_ROUND, 0u);
    io.ste(TE_E_AGENT_KILLS, 0u); io.ste(TE_E_ALLIES_KILLS, 0u); io.ste(TE_E_DEADS, 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k) io.ste(TE_E_LAST_ACTION + k, 0u);
    for (int j = P; j < D; ++j) { const V3 w = stage02_invader_position(c, env, j, episode, 0u); io.respawn(c, j, w); set_pos(j, w); }
    for (int q = 0; q < P; ++q) {
      const U4 r = env_rng(c, env, RNG_SPAWN_PURSUER, (uint32_t)q, 0, episode, 0);
      const V3 w = stage02_position(c.pursuer_spawn_radius, 0.0f, u01(r.x), u01(r.y), u01(r.z));
      io.respawn(c, q, w); set_pos(q, w);
    }
#pragma unroll
    for (int q = 0; q < PM; ++q) if (q < P) { mun[q] = max_munition_of(c, q); lf[q] = -c.cooldown_steps; }
    A = all_bits;
    io.ste(TE_E_SNAP_MASK, A);
    float d0 = 0.0f; bool any = false;   // closest invader of pursuer 0 on the fresh positions
#pragma unroll
    for (int j = 1; j < DM; ++j) {
      const float d = fdist(V3{px[0], py[0], pz[0]}, V3{px[j], py[j], pz[j]});
      const bool take = j >= P && j < D && (!any || d < d0);
      d0 = take ? d : d0; any = any || take;
    }
    io.stef(TE_E_PREV_SNAP_MIN, d0); io.stef(TE_E_LAST_DIST, d0);
    act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) ag[k] = 0.0f;
  }
  inertial_row_regs(c, row, px[0], py[0], pz[0], ag, mun[0], lf[0], step);
  write_rows_regs(p, o.obs, rows, lane, valid, env, row, act);
  {  // every armed slot flies as a dense wave outside the level4 family
    uint64_t dense = 0u;
    for (int s = 0; s < D; ++s) if (__ballot(valid && ((A >> s) & 1u))) dense |= (uint64_t)1 << s;
    if (lane == 0) { p.slot_mask[blockIdx.x] = dense; p.mixed_count[blockIdx.x] = 0u; p.live_mask[blockIdx.x] = dense; }
  }
}

// ---- stage01: PyflytL2EnviromentModifiedV2.step after the sim loop (pyflyt_level2_environment_modified_v2.py:137-211), the
// restatement of stage01_logic.  Slots: 0 = the RL pursuer, 1 = the hovering pursuer, 2 = the position-hold invader.
__global__ __launch_bounds__(64) void engage_stage01_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  constexpr int DM = 3;
  __shared__ float rows[64 * TE_OBS_INERTIAL_WORDS];
  const te_config& c = p.cfg;
  const int lane = threadIdx.x, env = blockIdx.x * 64 + lane;
  const bool valid = env < p.N;
  const EnvIO io(p, env);
  const GView g{p.dstate, p.estate, 3, p.Npad, env, 2};
  float px[DM], py[DM], pz[DM]; uint32_t A = 0u;
#pragma unroll
  for (int s = 0; s < DM; ++s) {
    px[s] = io.ldf(TE_D_OBS_POS, s); py[s] = io.ldf(TE_D_OBS_POS + 1, s); pz[s] = io.ldf(TE_D_OBS_POS + 2, s);
    A |= (io.ld(TE_D_ARMED, s) != 0u && valid ? 1u : 0u) << s;
  }
  int mun0 = (int)io.ld(TE_D_MUNITION, 0), lf0 = (int)io.ld(TE_D_LAST_FIRED, 0);
  float ag[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
  int step = (int)io.le(TE_E_STEP) + 1;
  const int max_step = (int)io.le(TE_E_MAX_STEP);
  int kills = (int)io.le(TE_E_AGENT_KILLS);
  uint32_t episode = io.le(TE_E_EPISODE);
  const float last = __uint_as_float(io.le(TE_E_LAST_DIST));
  float4 act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (valid) act = reinterpret_cast<const float4*>(actions)[env];
  if (valid) {
    io.stef(TE_E_LAST_ACTION + 0, act.x); io.stef(TE_E_LAST_ACTION + 1, act.y); io.stef(TE_E_LAST_ACTION + 2, act.z); io.stef(TE_E_LAST_ACTION + 3, act.w);
    io.ste(TE_E_STEP, (uint32_t)step);
  }
  const V3 pp{px[0], py[0], pz[0]}, pi{px[2], py[2], pz[2]};
  const float d = dist(pi, pp);
  float bonus = 0.0f, penalty = 0.0f;
  if (d < last) bonus += c.approach_bonus_gain * norm(V3{ag[3], ag[4], ag[5]});
  const bool caught = d < c.catch_distance;
  if (caught) bonus += 1000.0f;
  if (d > c.dome_radius) penalty += 1000.0f;
  const float reward = -d + bonus - penalty;
  const bool term = step > max_step || norm(pp) > c.dome_radius || norm(pi) > c.dome_radius;
  const bool to_terminal = valid && term && c.auto_reset;
  kills += caught ? 1 : 0;
  if (valid) {
    o.reward[env] = reward; o.done[env] = term ? 1 : 0;
    reinterpret_cast<int4*>(o.info)[env] = make_int4(kills, 0, 0, 0);
  }
  uint32_t cell[DM]; float rhat[DM];
  const uint32_t owners = own_sphere_regs<DM>(c, 3, px, py, pz, ag, A, cell, rhat);
  float row[TE_OBS_INERTIAL_WORDS];
  if (to_terminal) {
    inertial_row_regs(c, row, px[0], py[0], pz[0], ag, mun0, lf0, step);
    if (o.term.inertial) { for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) o.term.inertial[(size_t)env * TE_OBS_INERTIAL_WORDS + k] = row[k]; }
    if (o.term.last_action) reinterpret_cast<float4*>(o.term.last_action)[env] = act;
  }
  terminal_tiles(c, o.term.lidar, to_terminal, lane);
  if (valid) { patch_sphere_regs<DM>(c, to_terminal ? o.term.lidar : o.obs.lidar, env, 2, owners, cell, rhat); record_sphere_regs<DM>(o, p.Npad, env, to_terminal, owners, cell); }
  if (valid) {
    V3 p2 = pi;
    if (caught) {  // replace_invader_if_close (:147-154): rare, straight through the plane view (it reads the invader's motors and PID memories)
      p2 = stage01_cube(c, env, RNG_RESPAWN, 2, episode, (uint32_t)step);
      stage01_replace_invader(c, g, p2, episode, (uint32_t)step);
      io.ste(TE_E_AGENT_KILLS, (uint32_t)kills);
      px[2] = p2.x; py[2] = p2.y; pz[2] = p2.z;
    }
    io.stef(TE_E_LAST_DIST, dist(p2, pp));  // update_last_distance (:219-223)
  }
  if (to_terminal) {  // PyflytL2EnviromentModifiedV2.reset (:83-123): rare, through the plane view as well
    stage01_reset_env(c, g);
    episode += 1u; step = 0;
    const V3 n0 = stage01_cube(c, env, RNG_SPAWN_PURSUER, 0, episode, 0);
    px[0] = n0.x; py[0] = n0.y; pz[0] = n0.z;
    mun0 = 0; lf0 = (A & 1u) ? lf0 : -c.cooldown_steps;
    A = 7u;
    act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) ag[k] = 0.0f;
  }
  inertial_row_regs(c, row, px[0], py[0], pz[0], ag, mun0, lf0, step);
  write_rows_regs(p, o.obs, rows, lane, valid, env, row, act);
  {
    uint64_t dense = 0u;
    for (int s = 0; s < 3; ++s) if (__ballot(valid && ((A >> s) & 1u))) dense |= (uint64_t)1 << s;
    if (lane == 0) { p.slot_mask[blockIdx.x] = dense; p.mixed_count[blockIdx.x] = 0u; p.live_mask[blockIdx.x] = dense; }
  }
}

}  // namespace te
