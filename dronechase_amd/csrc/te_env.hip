// te_env.hip — gfx950 kernels + C ABI of the batched threat-engagement environment.
//
// One te_step = two launches on the caller's stream:
//   K1 substeps_kernel       one wavefront per workgroup.  256 fill waves stream the LIDAR background (ones);
//                            one DENSE drone wave per (slot, 64 consecutive envs), slot-major, for the slots armed
//                            in most envs of the chunk, and MIXED waves of 64 (env, slot) items for its sparsely
//                            armed slots (flight plan left by K2): armed lanes work out their set-point (an
//                            invader runs its navigator here) and fly the 16 physics sub-steps in registers
//                            (state read once, written once); a candidate wave with nothing to fly retires
//                            after one scalar load.
//   K2 engage_observe_kernel one 256-thread block per 64 envs, phases separated by barriers: stage the words the
//                            logic reads in LDS (one round of independent coalesced loads) -> precompute
//                            distances / masks / LIDAR cells, all threads -> wave 0, one lane per env:
//                            engagement, reward, termination, info, hit resolution, round / reset decision ->
//                            spawn phase, one (env, slot) per thread -> observation rows + the allies' commands
//                            and the flight plan of the next step -> patch the hit cells into the background.
// (te_step_stacked adds stacked_kernel, te_stacked.hpp; exp05 brackets te_step with te_observe_ally /
// te_set_ally_actions.)
// No MFMA: this is element-wise physics and byte streaming.
//
// Reference citations are file:line under the reference's src/ tree.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "te_logic.hpp"
#include "te_stacked.hpp"
#include "te_stackview.hpp"
#include "te_engage.hpp"
#include "te_engage_slots.hpp"

namespace te {

// ============================================================================================
// K1: sub-steps
// ============================================================================================
// One wavefront per workgroup, launched in this order:
//   [fill waves]  fill.n_fill_waves workgroups that only stream the LIDAR background (LIDARSpec.empty_sphere,
//                 angle_grid.py:95-104: all ones) with 16-byte stores, no loads: they sit next to the flying
//                 waves for the whole launch and keep HBM busy while the physics is latency-bound;
//   [drone waves] wave = slot * (Npad/64) + chunk, SLOT-major: the always-armed slots (agent, allies, the first
//                 invaders of the current round) reach the SIMDs first, the mostly disarmed slots retire after
//                 one load.  Chunk-major order let ~6 000 disarmed waves take the issue slots first.
// Measured on stage03, 65 536 envs (tools/k1_phase.py): chunk-major + per-wave fill 82-93 us; this order 53-71 us.
#ifndef TE_K1_BLOCK
#define TE_K1_BLOCK 64
#endif
// te_create: (env, slot) pairs up to which the sub-step launch carries noise-helper waves (TE_K1_HELP=0 / 1 overrides).  The helper pays while the
// launch has fewer flight waves than the chip has SIMDs (- 1.1 ... 2.0 us up to ~0.9 flights per SIMD, a tie at 1.3, + 3.5 us at 1.75 and above:
// profiles/r04_m_ab_noise_helper_v2.txt); te_create cannot know how many drones will be armed, so the default admits only shards on which EVERY slot
// armed is at most 1.5 flight waves per SIMD, where the helper roughly breaks even: n_envs x D <= 1.5 x 64 x 1 024 (stage01 up to 32 768 envs, stage03
// up to 8 936: the metric's 8-GPU shard, which flies 0.4-0.6 flights per SIMD in a random-action rollout).  A decision per launch from
// the flight plan was measured too (both waves of a block estimating the load from their chunk's plan): carrying both noise paths in one kernel cost
// the helped flights their gain (33.0 -> 32.8 us at 8 192 envs instead of 31.6) and the declined ones 7 us.
constexpr long long kHelpMaxPairs = 98304;
#ifdef TE_K1_WAVES
#define TE_K1_ATTR __attribute__((amdgpu_waves_per_eu(TE_K1_WAVES, TE_K1_WAVES)))
#else
#define TE_K1_ATTR
#endif
// Background job of one launch: the fill waves cover the whole buffer, grid-stride.
// the background is written once and not read by this launch: non-temporal ("nt") stores keep it from displacing the
// drone state in L2 (measured 58.7 -> 56.1 us per launch)
typedef float te_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t te_u4 __attribute__((ext_vector_type(4)));
#define TE_FILL_STORE(ptr) __builtin_nontemporal_store((te_f4){1.0f, 1.0f, 1.0f, 1.0f}, reinterpret_cast<te_f4*>(ptr))
// mode 0 / 1 / 2: stream ones over the whole buffer (pointer loop / scalar buffer loop / + s_setprio).  mode 3 (persistent observation,
// te_set_persistent_obs): the buffer still holds the previous observation; wave w = list * nchunks + chunk sets the cells that observation
// patched in output sphere `list` (= observer * 6 + sphere) of the chunk's 64 envs back to one — scattered stores that ride on the flights
// of the same launch (which leave HBM idle) instead of sitting in the stacked-observation launch's critical path.
struct FillJob { float* lidar; uint32_t total_quads; uint32_t n_fill_waves; uint32_t mode; const uint16_t* prev; uint32_t lists, list_rows, npad, n, tile_words; };

// what kamikaze_update() reads of the other drones, through the wave's buffer resource
struct NavView {
  const SlotLane& P; int Pn;
  // kamikaze_update() asks for a pursuer's TE_D_OBS_POS: serve the stable copy (see TE_X_REF)
  TE_DEV float gf(int w, int s) const { return P.lf_slot(TE_X_REF + (w - TE_D_OBS_POS), s); }
};

// One flight: the drone of slot `slot` of env `env` through the 16 sub-steps of this env.step.  In a DENSE wave `slot` is
// wave-uniform and the lanes are the 64 envs of a chunk; in a MIXED wave every lane carries its own (env, slot) item of
// the chunk's sparsely armed slots (Params::mixed_items), so `slot` is a per-lane value: the buffer addressing
// (SlotLane: per-lane byte offset + scalar plane offset) is the same for both.
// CES = cfg.control_every_substep (the reference's 240 Hz controller calls); false = PyFlyt-native 120 Hz
// EAGER (the small-shard kernels: HELP): a lone flight's start is a chain of memory round trips — is the candidate live (`gate`: a scalar load
// the caller has requested), is the lane armed, the state, and for an invader its FSM words and then the pursuers' reference positions — 3 to 5
// of them before the first sub-step, of a 17 us flight.  Here a valid lane requests ALL of it at once, armed or not, before anything is looked
// at (the invader's navigation then runs on registers: RegNav, at most kEagerP pursuers); a candidate with nothing to fly drops the requests.
// Large shards keep the lazy form: their 10 000 dead candidates must not touch vector memory (te_env.hip, substeps_kernel).
constexpr int kEagerP = 4;
struct RegNav {   // kamikaze_update's view of the pursuers' reference positions (TE_X_REF), already in registers
  float ref[3][kEagerP];
  TE_DEV float gf(int w, int s) const {
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < kEagerP; ++q) v = s == q ? ref[w - TE_D_OBS_POS][q] : v;
    return v;
  }
};
template <int FAMILY, bool NOISE, bool MIXED, bool CES = true, bool HELP = false, bool EAGER = false>
TE_DEV void fly(const Params& p, const float* __restrict__ actions, int slot, int env, bool valid, const float* nzbuf = nullptr, const volatile int* ready = nullptr, uint32_t gate = 1u) {
  const int D = p.D;
  const SlotLane P(p.dstate, p.estate, (uint32_t)D, (uint32_t)p.Npad, (uint32_t)slot, (uint32_t)env,
                   (uint32_t)(TE_DRONE_WORDS + TE_X_WORDS) * (uint32_t)D * (uint32_t)p.Npad, (uint32_t)TE_ENV_WORDS * (uint32_t)p.Npad);
  int armed = EAGER ? 0 : P.li(TE_D_ARMED);
  const bool want = EAGER ? valid : (valid && armed != 0);   // the lanes that request their state now
  const te_config& c = p.cfg;
  const bool mode7 = (FAMILY == FAM_STAGE01) && slot == 2;
  uint32_t nav_S = 0u; int nav_state = 0; V3 nav_me{0, 0, 0}; RegNav rnav;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int q = 0; q < kEagerP; ++q) rnav.ref[k][q] = 0.0f;

  // ---- every load of an armed lane is ISSUED before the background stores (in-order vmcnt: a load issued
  // after them could only be waited for together with them)
  float4 act = make_float4(0, 0, 0, 0);
  float cmd[4] = {0, 0, 0, 0};
  int nav_next = 0;
  float raw[29];
#pragma unroll
  for (int k = 0; k < 29; ++k) raw[k] = 0.0f;
  V3 pf{0, 0, 0}, pt{0, 0, 0};
  uint32_t episode = 0, step_index = 0;
  const bool scripted0 = FAMILY == FAM_LEVEL4 && all_scripted(c);  // Evaluation_Task / Level5DumbMultiObjectTask: pursuer 0 obeys the behaviour tree too
  if (want) {
    if (EAGER) armed = P.li(TE_D_ARMED);
    if (slot == 0 && !scripted0) act = reinterpret_cast<const float4*>(actions)[env];
    else if (FAMILY == FAM_LEVEL4) {
      if (slot < c.n_pursuers) {  // ally: command prepared by the previous engage/observe launch (or reset)
        cmd[0] = P.lf(TE_X_CMD + 0); cmd[1] = P.lf(TE_X_CMD + 1); cmd[3] = P.lf(TE_X_CMD + 2);
      } else if (EAGER) {         // invader: everything KamikazeNavigator.update may look at (evaluated below, on registers)
        nav_S = (uint32_t)P.lei(TE_E_SNAP_MASK); nav_state = P.li(TE_D_NAV_STATE);
        nav_me = V3{P.lf(TE_D_OBS_POS), P.lf(TE_D_OBS_POS + 1), P.lf(TE_D_OBS_POS + 2)};
#pragma unroll
        for (int q = 0; q < kEagerP; ++q)
          if (q < c.n_pursuers) { rnav.ref[0][q] = P.lf_slot(TE_X_REF + 0, q); rnav.ref[1][q] = P.lf_slot(TE_X_REF + 1, q); rnav.ref[2][q] = P.lf_slot(TE_X_REF + 2, q); }
      } else {                    // invader: KamikazeNavigator.update from the pursuers' last IMU positions
        const NavView nv{P, c.n_pursuers};
        const uint32_t S = (uint32_t)P.lei(TE_E_SNAP_MASK);
        const V3 me{P.lf(TE_D_OBS_POS), P.lf(TE_D_OBS_POS + 1), P.lf(TE_D_OBS_POS + 2)};
        float out[3];
        nav_next = kamikaze_update(c, nv, S, P.li(TE_D_NAV_STATE), me, out);
        cmd[0] = out[0]; cmd[1] = out[1]; cmd[3] = out[2];
      }
    } else {                          // stage01 / stage02: persistent set-points
#pragma unroll
      for (int k = 0; k < 4; ++k) cmd[k] = P.lf(TE_D_SETPOINT + k);
    }
#pragma unroll
    for (int k = 0; k < 29; ++k) raw[k] = P.lf(TE_D_POS + k);  // POS .. PID_ZV_E are words 0..28
    if (mode7) {
      pf = V3{P.lf(TE_D_PENDING), P.lf(TE_D_PENDING + 1), P.lf(TE_D_PENDING + 2)};
      pt = V3{P.lf(TE_D_PENDING + 3), P.lf(TE_D_PENDING + 4), P.lf(TE_D_PENDING + 5)};
    }
    episode = (uint32_t)P.lei(TE_E_EPISODE);
    // stage01 counts step_calls BEFORE the sim loop (pyflyt_level2_environment_modified_v2.py:128)
    step_index = (uint32_t)P.lei(TE_E_STEP) + (FAMILY == FAM_STAGE01 ? 1u : 0u);
  }
  if (EAGER) {
    if (!gate) return;   // wave-uniform: a candidate with nothing to fly (its sibling wave takes the same exit)
    if (HELP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the helper has cleared `ready`
  }
  const bool active = valid && armed != 0;
  if (!active) return;
  if (EAGER && FAMILY == FAM_LEVEL4 && !(slot == 0 && !scripted0) && slot >= c.n_pursuers) {
    float out[3];
    nav_next = kamikaze_update(c, rnav, nav_S, nav_state, nav_me, out);
    cmd[0] = out[0]; cmd[1] = out[1]; cmd[3] = out[2];
  }

  // ---- set-point for this env.step
  float sp[4];
  if (slot == 0 && !(FAMILY == FAM_LEVEL4 && all_scripted(c))) {  // RL agent: Quadcopter.drive (quadcopter.py:398-413)
    command_to_velocity(act.x, act.y, act.z, act.w, sp[0], sp[1], sp[3]);
    sp[2] = 0.0f;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) sp[k] = cmd[k];
    if (FAMILY == FAM_LEVEL4 && slot >= c.n_pursuers) P.si(TE_D_NAV_STATE, nav_next);
  }
  if (slot == 0 || FAMILY == FAM_LEVEL4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) P.sf(TE_D_SETPOINT + k, sp[k]);
  }
  Body b;
  b.pos = V3{raw[TE_D_POS], raw[TE_D_POS + 1], raw[TE_D_POS + 2]};
  b.q = Q4{raw[TE_D_QUAT], raw[TE_D_QUAT + 1], raw[TE_D_QUAT + 2], raw[TE_D_QUAT + 3]};
  {  // the loop's rotation_unit() assumes |q| = 1: true for every state this library writes, enforced for blobs
     // that came in through te_set_state
    const float inv = rsq(xfma(b.q.w, b.q.w, xfma(b.q.z, b.q.z, xfma(b.q.y, b.q.y, b.q.x * b.q.x))));
    b.q = Q4{b.q.x * inv, b.q.y * inv, b.q.z * inv, b.q.w * inv};
  }
  b.vel = V3{raw[TE_D_VEL], raw[TE_D_VEL + 1], raw[TE_D_VEL + 2]};
  b.wb = mulT(x_rotation(b.q), V3{raw[TE_D_OMEGA], raw[TE_D_OMEGA + 1], raw[TE_D_OMEGA + 2]});
#pragma unroll
  for (int k = 0; k < 4; ++k) b.thr[k] = raw[TE_D_THROTTLE + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { b.av_i[k] = raw[TE_D_PID_AV_I + k]; b.av_e[k] = raw[TE_D_PID_AV_E + k]; }
#pragma unroll
  for (int k = 0; k < 2; ++k) { b.lv_i[k] = raw[TE_D_PID_LV_I + k]; b.lv_e[k] = raw[TE_D_PID_LV_E + k]; }
  b.zv_i = raw[TE_D_PID_ZV_I]; b.zv_e = raw[TE_D_PID_ZV_E];

  // ---- fly.  The sub-step that captures the lagged IMU read is peeled out of the loop so the 12 observation
  // registers are not live (and conditionally written) across it.
  const int S = c.substeps;
  const Derived& kd = p.kd;
  constexpr bool kGround = FAMILY == FAM_LEVEL4;  // the opt-in ground plane (cfg.ground_contact) is compiled into the level4 family only
  // One Philox call feeds two consecutive sub-steps: words {x,y} go to the even one, {z,w} wait in two
  // registers for the odd one.
  const int n_plain = c.observe_lag ? S - 1 : S;
  uint32_t na = 0, nb = 0, held_a = 0, held_b = 0;
  // MIXED: (env, slot) ride through the loop in ONE register and are unpacked at each draw (a per-lane slot next to
  // env would cost the kernel its 72nd VGPR, i.e. a wave per SIMD); te_create caps n_envs per te_env below 2^24
  const uint32_t env_slot = (uint32_t)env | ((uint32_t)slot << 24);
  // HELP: rows published by the sibling wave so far, as last read (it runs ahead after the first sub-steps: a handful of polls per flight)
  int seen = 0;
#define TE_DRAW(s_)                                                                   \
  if (NOISE && HELP) {                                                                \
    while (seen <= (s_)) { seen = __builtin_amdgcn_readfirstlane(*ready); if (seen <= (s_)) __builtin_amdgcn_s_sleep(1); }   \
  }                                                                                   \
  if (NOISE && !HELP) {                                                               \
    if (((s_) & 1) == 0) {                                                            \
      uint32_t es_ = env_slot;                                                        \
      if (MIXED) asm volatile("" : "+v"(es_));                                        \
      const U4 bits = MIXED ? motor_noise_bits(c, (int)(es_ & 0xFFFFFFu), (int)(es_ >> 24), episode, step_index, (s_)) \
                            : motor_noise_bits(c, env, slot, episode, step_index, (s_));      \
      na = bits.x; nb = bits.y; held_a = bits.z; held_b = bits.w;                     \
    } else { na = held_a; nb = held_b; }                                              \
  }
  if (!CES) {
#pragma unroll
    for (int i = 0; i < 4; ++i) b.pwm[i] = 0.0f;   // sub-step 0 is always a controller sub-step
  }
#define NZROW(i_) (HELP ? reinterpret_cast<const volatile float4*>(nzbuf) + ((size_t)(i_) * 64 + (threadIdx.x & 63)) : nullptr)
  for (int s = 0; s < n_plain; ++s) {
    TE_DRAW(s)
    if (CES) {
      if (mode7) substep<true, false, NOISE, false, 1, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(s));  // wave-uniform: slot is per wave
      else substep<false, false, NOISE, kGround, 1, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(s));
    } else if (s % kd.ctrl_ratio == 0) {   // wave-uniform
      if (mode7) substep<true, false, NOISE, false, 0, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(s));
      else substep<false, false, NOISE, kGround, 0, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(s));
    } else {
      if (mode7) substep<true, false, NOISE, false, 2, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(s));
      else substep<false, false, NOISE, kGround, 2, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(s));
    }
  }
  if (c.observe_lag) {
    TE_DRAW(S - 1)
    if (CES) {
      if (mode7) substep<true, true, NOISE, false, 1, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(S - 1));
      else substep<false, true, NOISE, kGround, 1, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(S - 1));
    } else if ((S - 1) % kd.ctrl_ratio == 0) {
      if (mode7) substep<true, true, NOISE, false, 0, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(S - 1));
      else substep<false, true, NOISE, kGround, 0, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(S - 1));
    } else {
      if (mode7) substep<true, true, NOISE, false, 2, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(S - 1));
      else substep<false, true, NOISE, kGround, 2, HELP>(c, kd, b, sp, na, nb, pf, pt, NZROW(S - 1));
    }
  }
#undef NZROW
#undef TE_DRAW

  // ---- store
  const M3 R = x_rotation(b.q);
  if (!c.observe_lag) {  // IMU refreshed after the loop
    b.o_pos = b.pos; b.o_vel = mulT(R, b.vel); b.o_rate = b.wb; b.o_eul = euler_of(b.q);
  }
  const V3 ww = mul(R, b.wb);
  P.sf(TE_D_POS, b.pos.x); P.sf(TE_D_POS + 1, b.pos.y); P.sf(TE_D_POS + 2, b.pos.z);
  P.sf(TE_D_QUAT, b.q.x); P.sf(TE_D_QUAT + 1, b.q.y); P.sf(TE_D_QUAT + 2, b.q.z); P.sf(TE_D_QUAT + 3, b.q.w);
  P.sf(TE_D_VEL, b.vel.x); P.sf(TE_D_VEL + 1, b.vel.y); P.sf(TE_D_VEL + 2, b.vel.z);
  P.sf(TE_D_OMEGA, ww.x); P.sf(TE_D_OMEGA + 1, ww.y); P.sf(TE_D_OMEGA + 2, ww.z);
#pragma unroll
  for (int k = 0; k < 4; ++k) P.sf(TE_D_THROTTLE + k, b.thr[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { P.sf(TE_D_PID_AV_I + k, b.av_i[k]); P.sf(TE_D_PID_AV_E + k, b.av_e[k]); }
#pragma unroll
  for (int k = 0; k < 2; ++k) { P.sf(TE_D_PID_LV_I + k, b.lv_i[k]); P.sf(TE_D_PID_LV_E + k, b.lv_e[k]); }
  P.sf(TE_D_PID_ZV_I, b.zv_i); P.sf(TE_D_PID_ZV_E, b.zv_e);
  P.sf(TE_D_OBS_POS, b.o_pos.x); P.sf(TE_D_OBS_POS + 1, b.o_pos.y); P.sf(TE_D_OBS_POS + 2, b.o_pos.z);
  P.sf(TE_D_OBS_EULER, b.o_eul.x); P.sf(TE_D_OBS_EULER + 1, b.o_eul.y); P.sf(TE_D_OBS_EULER + 2, b.o_eul.z);
  P.sf(TE_D_OBS_VEL, b.o_vel.x); P.sf(TE_D_OBS_VEL + 1, b.o_vel.y); P.sf(TE_D_OBS_VEL + 2, b.o_vel.z);
  P.sf(TE_D_OBS_RATE, b.o_rate.x); P.sf(TE_D_OBS_RATE + 1, b.o_rate.y); P.sf(TE_D_OBS_RATE + 2, b.o_rate.z);
  if (mode7) {
#pragma unroll
    for (int k = 0; k < 6; ++k) P.sf(TE_D_PENDING + k, 0.0f);
  }
}

// Diagnostic builds (-DTE_DEBUG_STAMPS): every workgroup of the sub-step kernel leaves {start, end, HW_ID | XCC_ID << 32, kind} behind the
// engage kernel's records (tools/k1_waves.py: which SIMD flew how many waves, and when each finished).  kind: 0 idle candidate, 1 fill, 2 dense, 3 mixed
#ifdef TE_DEBUG_STAMPS
#define TE_K1_BASE(p) (64 + 16 * (size_t)((p).Npad / kEPB + 1))
#define TE_K1_BEGIN const unsigned long long k1_t0_ = __builtin_amdgcn_s_memrealtime();
#define TE_K1_END(kind)                                                                                               \
  do {                                                                                                                \
    if (p.dbg && threadIdx.x == 0) {                                                                                  \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
      unsigned long long* r_ = p.dbg + TE_K1_BASE(p) + 4 * (size_t)blockIdx.x;                                        \
      r_[0] = k1_t0_; r_[1] = __builtin_amdgcn_s_memrealtime();                                                       \
      r_[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
      r_[3] = (kind);                                                                                                 \
    }                                                                                                                 \
  } while (0)
#else
#define TE_K1_BEGIN
#define TE_K1_END(kind) do {} while (0)
#endif

// HELP: the sibling wave of a flight (small shards: fewer flights than SIMDs, every flight one lone wave).  The motor noise is the one part of a
// sub-step that does not depend on the drone's state — Philox4x32-7 on (env, slot, episode, step, sub-step) + Box-Muller: 19 % of a lone flight's
// time (noise off: 21.3 -> 17.5 us at 8 192 envs) — so this wave computes it for all sub-steps, two per Philox call like the flight would, and
// publishes row after row in LDS: nzbuf[sub-step][lane] = four normals, *ready = rows published.  Same functions, same operands, exact arithmetic:
// the flight flies on the same bits (tests/test_gpu_noise_helper.py).
template <int FAMILY, bool EAGER>
TE_DEV void noise_help(const Params& p, int slot, int env, bool valid, float* nzbuf, volatile int* ready, uint32_t gate) {
  const te_config& c = p.cfg;
  const int lane = threadIdx.x & 63;
  const SlotLane P(p.dstate, p.estate, (uint32_t)p.D, (uint32_t)p.Npad, (uint32_t)slot, (uint32_t)env,
                   (uint32_t)(TE_DRONE_WORDS + TE_X_WORDS) * (uint32_t)p.D * (uint32_t)p.Npad, (uint32_t)TE_ENV_WORDS * (uint32_t)p.Npad);
  const int armed = P.li(TE_D_ARMED);
  const uint32_t episode = (uint32_t)P.lei(TE_E_EPISODE);
  const uint32_t step_index = (uint32_t)P.lei(TE_E_STEP) + (FAMILY == FAM_STAGE01 ? 1u : 0u);
  if (EAGER) {   // (as the flight: requests first, then the candidate's flag; the lazy form has passed both in the kernel)
    if (!gate) return;
    if (lane == 0) *ready = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (__ballot(valid && armed != 0) == 0ull) return;   // the flight has nothing armed either and retires without asking
  const int S = c.substeps;
  for (int s = 0; s < S; s += 2) {
    const U4 bits = motor_noise_bits(c, env, slot, episode, step_index, s);
    float nz[4];
    motor_noise_from(bits.x, bits.y, p.kd.noise_m2ln2, nz);
    *reinterpret_cast<float4*>(nzbuf + ((size_t)s * 64 + lane) * 4) = make_float4(nz[0], nz[1], nz[2], nz[3]);
    if (s + 1 < S) {
      motor_noise_from(bits.z, bits.w, p.kd.noise_m2ln2, nz);
      *reinterpret_cast<float4*>(nzbuf + ((size_t)(s + 1) * 64 + lane) * 4) = make_float4(nz[0], nz[1], nz[2], nz[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *ready = min(s + 2, S);
  }
}

template <int FAMILY, bool NOISE, bool FILL, bool CES = true, bool HELP = false>
__global__ __launch_bounds__(HELP ? 128 : TE_K1_BLOCK) TE_K1_ATTR void substeps_kernel(Params p, const float* __restrict__ actions, FillJob fill) {
  TE_K1_BEGIN
  extern __shared__ float k1_lds[];   // HELP: [cfg.substeps][64][4] normals, then the `ready` word
  int wave = HELP ? (int)blockIdx.x : __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (unsigned)TE_K1_BLOCK + threadIdx.x) >> 6));
  const int role = HELP ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // 0 = the flight / fill wave, 1 = its noise helper
  volatile int* ready = reinterpret_cast<volatile int*>(k1_lds + (HELP ? (size_t)p.cfg.substeps * 256 : 0));
  const int lane = threadIdx.x & 63;
  const int D = p.D;
  const int nchunks = p.Npad >> 6;
  float4* fill_dst = reinterpret_cast<float4*>(fill.lidar);
  if (FILL) {
    if (HELP && role == 1 && wave < (int)fill.n_fill_waves) return;   // fill / erase blocks have no use for a helper
    if (wave < (int)fill.n_fill_waves && fill.mode == 3) {  // ---- erase wave (persistent observation)
      const uint32_t nch = fill.npad >> 6;
      const uint32_t list = (uint32_t)wave / nch, env = ((uint32_t)wave - list * nch) * 64u + (uint32_t)lane;
      if (env < fill.n) {
        const uint16_t* __restrict__ pv = fill.prev + (size_t)list * fill.list_rows * fill.npad + env;
        float* d0 = fill.lidar + ((size_t)env * fill.lists + list) * fill.tile_words;
        const int cnt = (int)pv[0];
        const bool time_plane = fill.tile_words > 2u * TE_LIDAR_CELLS;   // (cfg.lidar_channels == 2: no time plane)
        for (int pl = 0; pl < (time_plane ? 3 : 2); ++pl)   // plane by plane, like the patches
          for (int i = 0; i < cnt; ++i) d0[pl * TE_LIDAR_CELLS + pv[(size_t)(1 + i) * fill.npad]] = 1.0f;
      }
      TE_K1_END(1);
      return;
    }
    if (wave < (int)fill.n_fill_waves) {  // ---- fill wave
      // grid-stride: at any moment the fill waves write one contiguous n_fill_waves KB window, which the address
      // interleave spreads over every HBM channel
      if (fill.mode >= 1) {
        // Scalar loop: one buffer store per 1 KB block, the block offset in an SGPR.  The pointer form below costs three VALU
        // instructions per store (64-bit address, index, compare), and next to five or six flights on the same SIMD every one of them
        // queues behind the flights' VALU work: the background then finishes 10-20 us after the last flight (tools/k1_waves.py).
        if (fill.mode >= 2) __builtin_amdgcn_s_setprio(3);
        const uint32_t n_blocks = fill.total_quads >> 6;   // whole 1 KB blocks
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(fill.lidar, 0, (int)(n_blocks << 10), 0x00020000);
        const int voff = lane * 16;
        const te_u4 ones = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
        for (uint32_t b = (uint32_t)wave; b < n_blocks; b += fill.n_fill_waves)
          __builtin_amdgcn_raw_buffer_store_b128(ones, rs, voff, (int)(b << 10), 2 /* nt */);
        const uint32_t q = (n_blocks << 6) + (uint32_t)lane;   // the last partial block
        if ((uint32_t)wave == n_blocks % fill.n_fill_waves && q < fill.total_quads) TE_FILL_STORE(fill_dst + q);
        TE_K1_END(1);
        return;
      }
      const uint32_t stride = fill.n_fill_waves * 64u;
      for (uint32_t q = (uint32_t)wave * 64u + (uint32_t)lane; q < fill.total_quads; q += stride) TE_FILL_STORE(fill_dst + q);
      TE_K1_END(1);
      return;
    }
    wave -= (int)fill.n_fill_waves;
  }
  // Grid order after the fill waves: the dense candidates of the HEAD slots (agent, allies, first invader: armed in
  // nearly every env), then the mixed-wave candidates (whole flights: started last they would also finish last; started
  // first, their scalar look-ups delay the head flights by ~4 us), then the dense candidates of the remaining slots.
  const int n_head = min(D, p.cfg.n_pursuers + 1) * nchunks;
  const int n_mixed = (FAMILY == FAM_LEVEL4 && p.dense_min > 1) ? kMixedWaves * nchunks : 0;
  if (wave >= n_head && wave < n_head + n_mixed) {  // ---- mixed wave m of a chunk: 64 (env, slot) items of its sparsely armed slots
    wave -= n_head;
    const int m = wave / nchunks, chunk = wave - m * nchunks;
    const int count = __builtin_amdgcn_readfirstlane((int)((const uint32_t* __restrict__)p.mixed_count)[chunk]);
    if (m * 64 >= count) { TE_K1_END(0); return; }
    const int i = m * 64 + lane;
    const bool valid = i < count;
    const uint32_t item = valid ? (uint32_t)p.mixed_items[(size_t)chunk * kMixedCap + i] : 0u;
    if (HELP && role == 1) { noise_help<FAMILY, true>(p, (int)(item >> 8), chunk * 64 + (int)(item & 63u), valid, k1_lds, ready, 1u); return; }
    fly<FAMILY, NOISE, true, CES, HELP, HELP>(p, actions, (int)(item >> 8), chunk * 64 + (int)(item & 63u), valid, k1_lds, ready);
    TE_K1_END(3);
    return;
  }
  if (wave >= n_head) wave -= n_mixed;
  const int slot = wave / nchunks;
  const int chunk = wave - slot * nchunks;
  if (slot >= D) { TE_K1_END(0); return; }
  // Does this (slot, chunk) fly as a dense wave?  One wave-uniform SCALAR load: the ~8 000 idle waves of a launch used
  // to wait for a vector flag load (and issue a background store) behind the fill waves' stores: +15 us per launch.
  const uint32_t* __restrict__ sm32 = reinterpret_cast<const uint32_t*>(p.slot_mask);
  const uint32_t chunk_mask = __builtin_amdgcn_readfirstlane(sm32[2 * chunk + (slot >> 5)]);   // the 32-bit half that holds this slot's bit
  const int env = chunk * 64 + lane;  // planes are padded to Npad: lanes beyond N still read in bounds
  if (HELP && FAMILY == FAM_LEVEL4) {   // small shards: the flag is looked at behind the flight's (and the helper's) requests (fly<..., EAGER>)
    const uint32_t gate = (chunk_mask >> (slot & 31)) & 1u;
    if (role == 1) { noise_help<FAMILY, true>(p, slot, env, env < p.N, k1_lds, ready, gate); return; }
    fly<FAMILY, NOISE, false, CES, true, true>(p, actions, slot, env, env < p.N, k1_lds, ready, gate);
    TE_K1_END(2);
    return;
  }
  if (!((chunk_mask >> (slot & 31)) & 1u)) { TE_K1_END(0); return; }
  if (HELP) {   // (stage01 / stage02: two round trips in front of a flight, and the eager form measured 0.4 us slower there)
    if (role == 1 && lane == 0) *ready = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (role == 1) { noise_help<FAMILY, false>(p, slot, env, env < p.N, k1_lds, ready, 1u); return; }
  }
  fly<FAMILY, NOISE, false, CES, HELP>(p, actions, slot, env, env < p.N, k1_lds, ready);
  TE_K1_END(2);
}

// ============================================================================================
// K2: engagement / reward / termination / waves / auto-reset / observation
// ============================================================================================
template <int FAMILY>
__global__ __launch_bounds__(256) void reset_kernel(Params p, const uint8_t* __restrict__ mask) {
  int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= p.N) return;
  if (mask && !mask[env]) return;
  GView v{p.dstate, p.estate, p.D, p.Npad, env, p.cfg.n_pursuers};
  reset_env<FAMILY>(p.cfg, v);
  if (p.ring)  // base_lidar.py:62-66: the step-0 broadcast resets every LIDAR buffer
    for (int k = 0; k < p.cfg.n_pursuers * TE_RING_DEPTH; ++k) p.ring[((size_t)env * p.cfg.n_pursuers * TE_RING_DEPTH + k) * (size_t)p.entry_words] = 0u;
}
// te_observe_stacked: the snapshot planes of the CURRENT state (the engage/observe kernel writes them during a step)
__global__ __launch_bounds__(256) void snapshot_kernel(Params p) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= p.N) return;
  const GView v{p.dstate, p.estate, p.D, p.Npad, env, p.cfg.n_pursuers};
  const SnapRows sr{p.D, p.cfg.n_pursuers};
  for (int k = 0; k < 3; ++k) {
    for (int s = 0; s < p.D; ++s) p.snap[(size_t)(sr.pos() + k * p.D + s) * p.Npad + env] = (uint32_t)v.gi(TE_D_OBS_POS + k, s);
    for (int s = 0; s < sr.P; ++s) p.snap[(size_t)(sr.euler() + k * sr.P + s) * p.Npad + env] = (uint32_t)v.gi(TE_D_OBS_EULER + k, s);
  }
  p.snap[(size_t)sr.armed() * p.Npad + env] = armed_mask(v);
  p.snap[(size_t)sr.step() * p.Npad + env] = (uint32_t)v.egi(TE_E_STEP);
  p.snap[(size_t)sr.episode() * p.Npad + env] = (uint32_t)v.egi(TE_E_EPISODE);
  p.snap[(size_t)sr.done() * p.Npad + env] = 0u;
  uint32_t hi = 0u;
  for (int s = 32; s < p.D; ++s) hi |= (v.gi(TE_D_ARMED, s) ? 1u : 0u) << (s - 32);
  p.snap[(size_t)sr.armed_hi() * p.Npad + env] = hi;
}
// The flight plan of a chunk for the next sub-step launch, by ONE wave whose lanes are the chunk's 64 envs, slot by slot:
// slots armed in >= kDenseMin envs (and the agent's) fly as dense waves (bit in slot_mask), the armed (env, slot) pairs of
// the other slots go to the mixed list in slot order.  `a` = this lane's env has drone s armed (false for lanes >= nvalid).
template <int FAMILY>
TE_DEV void plan_slot(uint16_t* __restrict__ items, int dense_min, int lane, int s, bool a, uint64_t& dense, int& n, uint64_t& live) {
  const unsigned long long b = __ballot(a);
  const int cnt = __popcll(b);
  if (cnt == 0) return;
  live |= (uint64_t)1 << s;   // armed somewhere in the chunk: the engage kernel requests this slot's rows (Params::live_mask)
  if (FAMILY != FAM_LEVEL4 || s == 0 || cnt >= dense_min || n + cnt > kMixedCap) { dense |= (uint64_t)1 << s; return; }
  // rank of this lane among the armed ones: v_mbcnt counts the set bits of b below the lane
  if (a) items[n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = (uint16_t)(lane | (s << 8));
  n += cnt;
}
TE_DEV void plan_done(const Params& p, int chunk, int lane, uint64_t dense, int n, uint64_t live) {
  if (lane == 0) { p.slot_mask[chunk] = dense; p.mixed_count[chunk] = (uint32_t)n; p.live_mask[chunk] = live; }
}
// rebuild slot_mask from the armed planes (after te_create / te_reset / te_set_state; during a rollout the
// engage/observe kernel maintains it): one 64-thread block per chunk of 64 envs
template <int FAMILY>
__global__ __launch_bounds__(64) void census_kernel(Params p) {
  const int chunk = blockIdx.x, l = threadIdx.x, env = chunk * 64 + l;
  uint64_t dense = 0u, live = 0u; int n = 0;
  uint16_t* items = p.mixed_items + (size_t)chunk * kMixedCap;
  for (int s = 0; s < p.D; ++s) plan_slot<FAMILY>(items, p.dense_min, l, s, env < p.N && p.dstate[((size_t)TE_D_ARMED * p.D + s) * p.Npad + env] != 0u, dense, n, live);
  plan_done(p, chunk, l, dense, n, live);
}
// recompute the pending scripted commands from a freshly loaded state blob (te_set_state)
__global__ __launch_bounds__(256) void prepare_commands_kernel(Params p) {
  int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= p.N) return;
  GView v{p.dstate, p.estate, p.D, p.Npad, env, p.cfg.n_pursuers};
  prepare_level4_commands(p.cfg, v);
}

// One round of independent, coalesced loads: every word the per-env logic reads -> LDS rows.
// Loads are issued in batches of 16 per wave BEFORE any of them is consumed, so a block pays a couple of
// memory latencies here instead of one per row.
// plane of staged row `row` as a word offset (bit 31: env record instead of drone state): host side, once per te_env
static uint32_t staged_row_offset(const Params& p, const Rows& r, int row) {
  const int D = p.D;
  size_t w;
  if (row < r.munition()) {  // OBS_POS (3*D rows) then ARMED (D rows): plane = base word + row / D
    const int q = row / D, s = row - q * D;
    w = ((size_t)(q < 3 ? TE_D_OBS_POS + q : TE_D_ARMED) * D + s) * p.Npad;
  } else if (row < r.agent()) {  // MUNITION, LAST_FIRED of the P pursuers
    const int k = row - r.munition();
    w = ((size_t)(k < r.P ? TE_D_MUNITION : TE_D_LAST_FIRED) * D + (k < r.P ? k : k - r.P)) * p.Npad;
  } else if (row < r.env()) w = ((size_t)(TE_D_OBS_EULER + (row - r.agent())) * D) * p.Npad;
  else return 0x80000000u | (uint32_t)((size_t)(row - r.env()) * p.Npad);
  return (uint32_t)w;
}
// One round of independent loads: row -> (scalar table lookup) -> buffer load with the plane as scalar offset and ONE
// per-lane byte offset.  (Computing each row's plane with per-lane integer arithmetic cost ~48 VALU instructions per
// load: 85 % of all VALU instructions of the kernel.)
TE_DEV void stage_block(const Params& p, uint32_t* sm, const Rows& r, int env0) {
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.dstate, 0, (int)((uint32_t)(TE_DRONE_WORDS + TE_X_WORDS) * (uint32_t)p.D * (uint32_t)p.Npad * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(p.estate, 0, (int)((uint32_t)TE_ENV_WORDS * (uint32_t)p.Npad * 4u), 0x00020000);
  const int voff = (env0 + lane) * 4;  // planes are padded to Npad (multiple of 64): always in bounds
  const uint32_t* __restrict__ tab = p.stage_tab;
  const int n = r.staged();
  constexpr int B = 32;  // 4 waves x 32 >= 112 rows (D = 11): the whole block is staged in ONE round of loads
  for (int base = wave; base < n; base += B * nw) {
    uint32_t vals[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const int row = base + k * nw;
      vals[k] = 0u;
      if (row < n) {
        const uint32_t t = __builtin_amdgcn_readfirstlane(tab[row]);
        vals[k] = (t >> 31) ? __builtin_amdgcn_raw_buffer_load_b32(re, voff, (int)((t & 0x7FFFFFFFu) << 2), 0)
                            : __builtin_amdgcn_raw_buffer_load_b32(rd, voff, (int)(t << 2), 0);
      }
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const int row = base + k * nw;
      if (row < n) sm[row * kEPB + lane] = vals[k];
    }
  }
}

// THREADS = 256, or 512 when the block's LDS leaves room for only two blocks per CU (many drones per env, level5): eight
// waves per block keep four waves per SIMD there; every loop strides by blockDim.x.
template <int FAMILY, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void engage_observe_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  const Rows r{p.D, p.cfg.n_pursuers};
  const int env0 = blockIdx.x * kEPB;
  const int nvalid = min(kEPB, p.N - env0);
  // The [N,3,13,26] LIDAR buffer has already been filled with ones by the sub-step kernel's waves
  // (FillJob): only the hit cells are patched here.
  TE_STAMP(p, 500, 0);
  if (threadIdx.x < kEPB) sm[r.task() * kEPB + threadIdx.x] = 0u;
  // the logic lane's action: requested now, consumed two barriers later (a load issued inside the logic phase was a
  // 2 us stall at its very top)
  float4 my_action = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (threadIdx.x < kEPB && (int)threadIdx.x < nvalid) my_action = reinterpret_cast<const float4*>(actions)[env0 + threadIdx.x];
  stage_block(p, sm, r, env0);
  __syncthreads();
  TE_STAMP(p, 500, 1);
  precompute_block(p.cfg, sm, r);
  __syncthreads();
  TE_STAMP(p, 500, 2);
  if (threadIdx.x < kEPB && (int)threadIdx.x < nvalid) {
    const int lane = threadIdx.x;
    SView v{GView{p.dstate, p.estate, p.D, p.Npad, env0 + lane, r.P}, sm, lane, r, p.D, r.P, env0 + lane, true};
    const float4 a = my_action;
    if (FAMILY == FAM_STAGE01) stage01_logic(p.cfg, v, a, o);
    else if (FAMILY == FAM_STAGE02) stage02_logic(p.cfg, v, a, o);
    else level4_logic(p.cfg, v, a, o);
  }
  TE_STAMP(p, 500, 3);
  __syncthreads();
  TE_STAMP(p, 500, 4);
  if (FAMILY == FAM_LEVEL4 && p.snap) {
    // level5: leave what this step's stacked observation may look at for stacked_kernel, BEFORE anything respawns
    const SnapRows sr{p.D, r.P};
    for (int it = threadIdx.x; it < kEPB * sr.total(); it += blockDim.x) {
      const int l = it & (kEPB - 1), w = it / kEPB;
      uint32_t v;
      if (w < 3 * p.D) v = sm[(r.obs_pos() + w) * kEPB + l];
      else if (w < sr.armed()) { const int k = (w - 3 * p.D) / r.P, s = (w - 3 * p.D) - k * r.P; v = p.dstate[((size_t)(TE_D_OBS_EULER + k) * p.D + s) * p.Npad + env0 + l]; }
      else if (w == sr.armed()) v = sm[r.anow() * kEPB + l];
      else if (w == sr.step()) v = sm[r.sstep() * kEPB + l];
      else if (w == sr.episode()) v = sm[r.sepis() * kEPB + l];
      else if (w == sr.done()) v = sm[r.done() * kEPB + l];
      else v = 0u;   // armed_hi: this kernel serves at most 32 drones per env
      p.snap[(size_t)w * p.Npad + env0 + l] = v;
    }
    __syncthreads();  // the spawn phase overwrites the obs_pos rows
  }
  if (FAMILY == FAM_LEVEL4) {
    // spawn phase: the slots of every env that starts a new round or auto-resets, one (env, slot) per thread
    // iteration (slot-major: consecutive threads = consecutive envs).  Rare per env, but with 64 envs per
    // block some block needs it almost every step, and left to the env's own lane it doubles that block's
    // critical path (D Philox draws + trigonometry + ~40 stores each, serially).
    const uint32_t my_task = threadIdx.x < kEPB ? sm[r.task() * kEPB + threadIdx.x] : 0u;
    if (__syncthreads_or(my_task != 0u)) {
      // compact the flagged envs first (their lane numbers go to the front of the, now free, `lcell` row), so that the
      // (env, slot) items spread over all 256 threads even when only a few envs of the block respawn
      if (threadIdx.x < kEPB) {
        const unsigned long long flagged = __ballot(my_task != 0u && (int)threadIdx.x < nvalid);
        if (my_task != 0u && (int)threadIdx.x < nvalid)
          sm[r.lcell() * kEPB + __popcll(flagged & ((1ull << threadIdx.x) - 1ull))] = threadIdx.x;
        if (threadIdx.x == 0) sm[r.lrhat() * kEPB] = (uint32_t)__popcll(flagged);  // slot 0 of lcell / lrhat is never a target: both rows are free
      }
      __syncthreads();
      const int n_items = (int)sm[r.lrhat() * kEPB] * p.D;
      for (int it = threadIdx.x; it < n_items; it += blockDim.x) {
        const int s = it % p.D, l = (int)sm[r.lcell() * kEPB + it / p.D];
        const uint32_t t = sm[r.task() * kEPB + l];
        SView v{GView{p.dstate, p.estate, p.D, p.Npad, env0 + l, r.P}, sm, l, r, p.D, r.P, env0 + l, false};
        level4_spawn_slot(p.cfg, v, s, (int)(t & 0xFFu), (uint32_t)v.egi(TE_E_EPISODE), (t >> 8) != 0u);
      }
      __syncthreads();
    }
    // epilogue split by wave: wave 3 prepares the allies' commands of the next step, one (env, ally) item per
    // thread iteration, while waves 0..2 write the inertial / last_action rows
    if (threadIdx.x < 192) emit_rows(p.cfg, sm, r, o.obs, env0, nvalid, threadIdx.x, 192);
    else {
      const int nprep = (int)blockDim.x - 192;  // one wave of a 256-thread block, five of a 512-thread one
      for (int it = threadIdx.x - 192; it < kEPB * r.P; it += nprep) {  // the invaders' reference of every pursuer (post-spawn)
        const int l = it & (kEPB - 1), s = it / kEPB;
        if (l >= nvalid) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) p.dstate[((size_t)(TE_X_REF + k) * p.D + s) * p.Npad + env0 + l] = sm[(r.obs_pos() + k * p.D + s) * kEPB + l];
      }
      const int first = p.cfg.evaluation ? 0 : 1;  // Evaluation_Task scripts pursuer 0 as well
      for (int it = threadIdx.x - 192; it < kEPB * (r.P - first); it += nprep) {
        const int l = it & (kEPB - 1), s = first + it / kEPB;
        if (l >= nvalid) continue;
        SView v{GView{p.dstate, p.estate, p.D, p.Npad, env0 + l, r.P}, sm, l, r, p.D, r.P, env0 + l, sm[r.prevalid() * kEPB + l] != 0u};
        prepare_slot(p.cfg, v, s);
      }
    }
  } else {
    emit_rows(p.cfg, sm, r, o.obs, env0, nvalid, threadIdx.x, blockDim.x);
  }
  if (threadIdx.x < kEPB) {  // what the next sub-step launch has to fly for this chunk (post-spawn flags)
    uint64_t dense = 0u, live = 0u; int n = 0;
    uint16_t* items = p.mixed_items + (size_t)blockIdx.x * kMixedCap;
    for (int s = 0; s < p.D; ++s)
      plan_slot<FAMILY>(items, p.dense_min, (int)threadIdx.x, s, (int)threadIdx.x < nvalid && sm[(r.armed() + s) * kEPB + threadIdx.x] != 0u, dense, n, live);
    plan_done(p, (int)blockIdx.x, (int)threadIdx.x, dense, n, live);
  }
  TE_STAMP(p, 500, 5);
  // terminal tiles of auto-reset envs (rare, block-uniform test): ones, drained, then patched
  const bool lane_done = threadIdx.x < kEPB && sm[r.done() * kEPB + threadIdx.x] != 0u;
  if (o.term.lidar && __syncthreads_or(lane_done ? 1 : 0)) {
    stream_terminal_ones(p.cfg, sm, r, o.term.lidar, env0, nvalid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  TE_STAMP(p, 500, 6);
  patch_hits(p.cfg, sm, r, o.obs.lidar, o.term.lidar, env0, nvalid);
  TE_STAMP(p, 500, 7);
}

// Background of the LIDAR observation: all ones (LIDARSpec.empty_sphere, angle_grid.py:95-104).  Pure
// 16-byte store stream; runs on the side stream concurrently with the VALU-bound sub-step kernel.
__global__ __launch_bounds__(256) void fill_ones_kernel(float* __restrict__ dst, size_t n_floats) {
  const float4 ones = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
  const size_t quads = n_floats >> 2, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < quads; i += stride) reinterpret_cast<float4*>(dst)[i] = ones;
  for (size_t i = (quads << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_floats; i += stride) dst[i] = 1.0f;
}

// exp05: Exp05_vFinal_Task.compute_lw_observation (exp05_vFinal_task.py:265-292) of pursuer `me` = 1.  One workgroup per
// chunk of 64 envs, straight from the state planes (coalesced over envs): (env, drone) items compute the LIDAR cell and
// range of every other armed drone seen from the ally's IMU attitude into LDS, all threads stream the chunk's tile
// (ones), then the owner of each cell is patched in.  Same rules as the agent's sphere: closer wins in slot order, empty
// right after a reset.
__global__ __launch_bounds__(256) void observe_ally_kernel(Params p, int me, float* __restrict__ lidar, float* __restrict__ inertial,
                                                           float* __restrict__ last_action, uint8_t* __restrict__ active) {
  __shared__ uint32_t s_cell[kMaxD * kEPB];
  __shared__ float s_rhat[kMaxD * kEPB];
  const te_config& c = p.cfg;
  const int D = p.D;
  const int env0 = blockIdx.x * kEPB, nvalid = min(kEPB, p.N - env0);
  const int l = threadIdx.x & (kEPB - 1), w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool valid = l < nvalid;
  const GView v{p.dstate, p.estate, D, p.Npad, env0 + l, c.n_pursuers};   // planes are padded to Npad: in bounds for every lane
  const int step = v.egi(TE_E_STEP);
  const bool sees = lidar && valid && step != 0;
  // wave roles: the last wave only streams the tile (a wave that has stores in flight waits for them before any later
  // load returns: in-order vmcnt), the others compute the features meanwhile
  const int nfeat = nw - 1;
  if (sees && w < nfeat) {
    const Q4 q = quat_of_euler(V3{v.gf(TE_D_OBS_EULER, me), v.gf(TE_D_OBS_EULER + 1, me), v.gf(TE_D_OBS_EULER + 2, me)});
    const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const M3 R = rotation(Q4{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2});
    const V3 own = obs_pos(v, me);
    for (int j = w; j < D; j += nfeat) {
      uint32_t cell = 0xFFFFFFFFu; float rhat = 1.0f;
      if (j != me && v.gi(TE_D_ARMED, j)) { int cj; lidar_cell(c, mul(R, sub(obs_pos(v, j), own)), cj, rhat); cell = (uint32_t)cj; }
      s_cell[j * kEPB + l] = cell; s_rhat[j * kEPB + l] = rhat;
    }
  }
  if (lidar && w == nfeat) {  // the chunk's tile: 64 * 1014 floats, 16-byte aligned; non-temporal like the sub-step kernel's background
    float* tile = lidar + (size_t)env0 * lidar_words(c);
    const int total = nvalid * lidar_words(c), quads = total >> 2;
    for (int qi = l; qi < quads; qi += kEPB) TE_FILL_STORE(reinterpret_cast<float4*>(tile) + qi);
    for (int f = (quads << 2) + l; f < total; f += kEPB) tile[f] = 1.0f;
  }
  if (w == 0 && valid) {
    const int env = env0 + l;
    if (active) active[env] = v.gi(TE_D_ARMED, me) ? 1 : 0;
    if (last_action)
      reinterpret_cast<float4*>(last_action)[env] = make_float4(v.gf(TE_D_ALLY_ACTION, me), v.gf(TE_D_ALLY_ACTION + 1, me),
                                                                 v.gf(TE_D_ALLY_ACTION + 2, me), v.gf(TE_D_ALLY_ACTION + 3, me));
    if (inertial) {
      float in[TE_OBS_INERTIAL_WORDS];
      inertial_obs(c, v, step, in, me);
#pragma unroll
      for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) inertial[(size_t)env * TE_OBS_INERTIAL_WORDS + k] = in[k];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ones must have landed before any patch
  __syncthreads();
  if (!sees) return;
  // owner of a cell = smallest range, the earlier slot on ties (strict '<' in slot order, lidar_math.py:262-311); a
  // feature clipped to 1.0 never enters an empty cell
  float* base = lidar + (size_t)(env0 + l) * lidar_words(c);
  for (int j = w; j < D; j += nw) {
    const uint32_t cell = s_cell[j * kEPB + l];
    const float rhat = s_rhat[j * kEPB + l];
    if (cell == 0xFFFFFFFFu || !(rhat < 1.0f)) continue;
    bool owner = true;
    for (int k = 0; k < D; ++k) {
      if (k == j || s_cell[k * kEPB + l] != cell) continue;
      const float rk = s_rhat[k * kEPB + l];
      if (rk < rhat || (rk == rhat && k < j)) owner = false;
    }
    if (!owner) continue;
    base[cell] = rhat;
    base[TE_LIDAR_CELLS + cell] = (float)(j < c.n_pursuers ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;
    if (c.lidar_channels != 2) base[2 * TE_LIDAR_CELLS + cell] = 0.1f;
  }
}
// The same observation as two launches for full-size batches (48 us instead of 64 at 65 536 envs):
//   ally_view_kernel   single-wave workgroups: `n_fill` of them stream the background in the sub-step kernel's fill-wave
//                      shape while one wave per chunk (lane = env) computes cells / ranges / owners and leaves them as
//                      (cell | type << 16, r_hat) planes in a scratch buffer, and writes the inertial / last-action rows;
//   ally_patch_kernel  one thread per (env, drone): the owners' three floats into the finished background.
__global__ __launch_bounds__(64) void ally_view_kernel(Params p, int me, float* __restrict__ lidar, uint32_t quads, uint32_t n_fill,
                                                       float* __restrict__ inertial, float* __restrict__ last_action,
                                                       uint8_t* __restrict__ active, uint32_t* __restrict__ scratch) {
  __shared__ uint32_t s_cell[kMaxD * kEPB];
  __shared__ float s_rhat[kMaxD * kEPB];
  const int l = threadIdx.x;
  if (blockIdx.x < n_fill) {
    const uint32_t stride = n_fill * 64u;
    for (uint32_t q = blockIdx.x * 64u + (uint32_t)l; q < quads; q += stride) TE_FILL_STORE(reinterpret_cast<float4*>(lidar) + q);
    return;
  }
  const te_config& c = p.cfg;
  const int D = p.D;
  const int env = (int)(blockIdx.x - n_fill) * kEPB + l;
  const bool valid = env < p.N;
  const GView v{p.dstate, p.estate, D, p.Npad, env, c.n_pursuers};   // planes are padded to Npad: in bounds for every lane
  const int step = v.egi(TE_E_STEP);
  const bool sees = valid && step != 0;
  if (sees) {
    const Q4 q = quat_of_euler(V3{v.gf(TE_D_OBS_EULER, me), v.gf(TE_D_OBS_EULER + 1, me), v.gf(TE_D_OBS_EULER + 2, me)});
    const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const M3 R = rotation(Q4{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2});
    const V3 own = obs_pos(v, me);
    for (int j = 0; j < D; ++j) {
      uint32_t cell = 0xFFFFFFFFu; float rhat = 1.0f;
      if (j != me && v.gi(TE_D_ARMED, j)) { int cj; lidar_cell(c, mul(R, sub(obs_pos(v, j), own)), cj, rhat); cell = (uint32_t)cj; }
      s_cell[j * kEPB + l] = cell; s_rhat[j * kEPB + l] = rhat;
    }
  }
  if (valid) {
    if (active) active[env] = v.gi(TE_D_ARMED, me) ? 1 : 0;
    if (last_action)
      reinterpret_cast<float4*>(last_action)[env] = make_float4(v.gf(TE_D_ALLY_ACTION, me), v.gf(TE_D_ALLY_ACTION + 1, me),
                                                                 v.gf(TE_D_ALLY_ACTION + 2, me), v.gf(TE_D_ALLY_ACTION + 3, me));
    if (inertial) {
      float in[TE_OBS_INERTIAL_WORDS];
      inertial_obs(c, v, step, in, me);
#pragma unroll
      for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) inertial[(size_t)env * TE_OBS_INERTIAL_WORDS + k] = in[k];
    }
  }
  // owner of a cell = smallest range, the earlier slot on ties; a feature clipped to 1.0 never enters an empty cell
  // (each lane reads back only what it wrote: no barrier)
  for (int j = 0; j < D; ++j) {
    uint32_t out = 0xFFFFFFFFu; float rh = 1.0f;
    if (sees) {
      const uint32_t cell = s_cell[j * kEPB + l];
      const float rhat = s_rhat[j * kEPB + l];
      if (cell != 0xFFFFFFFFu && rhat < 1.0f) {
        bool owner = true;
        for (int k = 0; k < D; ++k) {
          if (k == j || s_cell[k * kEPB + l] != cell) continue;
          const float rk = s_rhat[k * kEPB + l];
          if (rk < rhat || (rk == rhat && k < j)) owner = false;
        }
        if (owner) { out = cell | ((uint32_t)(j < c.n_pursuers ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) << 16); rh = rhat; }
      }
    }
    scratch[(size_t)(2 * j) * p.Npad + env] = out; scratch[(size_t)(2 * j + 1) * p.Npad + env] = __float_as_uint(rh);
  }
}
__global__ __launch_bounds__(256) void ally_patch_kernel(Params p, float* __restrict__ lidar, const uint32_t* __restrict__ scratch) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)p.D * p.Npad) return;
  const int j = (int)(i / p.Npad), env = (int)(i - (size_t)j * p.Npad);
  if (env >= p.N) return;
  const uint32_t w = scratch[(size_t)(2 * j) * p.Npad + env];
  if (w == 0xFFFFFFFFu) return;
  float* d = lidar + (size_t)env * lidar_words(p.cfg) + (w & 0xFFFFu);
  d[0] = __uint_as_float(scratch[(size_t)(2 * j + 1) * p.Npad + env]); d[TE_LIDAR_CELLS] = (float)(w >> 16) / 5.0f;
  if (p.cfg.lidar_channels != 2) d[2 * TE_LIDAR_CELLS] = 0.1f;
}
// exp05: pursuer.drive(action) of drive_lw_rl_agent (exp05_vFinal_task.py:255-260; quadcopter.py:379-413) for armed allies
__global__ __launch_bounds__(256) void set_ally_actions_kernel(Params p, int me, const float* __restrict__ actions) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= p.N) return;
  const GView v{p.dstate, p.estate, p.D, p.Npad, env, p.cfg.n_pursuers};
  if (!v.gi(TE_D_ARMED, me)) return;
  const float4 a = reinterpret_cast<const float4*>(actions)[env];
  float vx, vy, vz;
  command_to_velocity(a.x, a.y, a.z, a.w, vx, vy, vz);
  v.sf(TE_X_CMD + 0, me, vx); v.sf(TE_X_CMD + 1, me, vy); v.sf(TE_X_CMD + 2, me, vz);
  v.sf(TE_D_SETPOINT + 0, me, vx); v.sf(TE_D_SETPOINT + 1, me, vy); v.sf(TE_D_SETPOINT + 2, me, 0.0f); v.sf(TE_D_SETPOINT + 3, me, vz);
  v.sf(TE_D_ALLY_ACTION + 0, me, a.x); v.sf(TE_D_ALLY_ACTION + 1, me, a.y); v.sf(TE_D_ALLY_ACTION + 2, me, a.z); v.sf(TE_D_ALLY_ACTION + 3, me, a.w);
}

// te_observe: current observation without stepping
__global__ __launch_bounds__(256) void observe_kernel(Params p, ObsOut o) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  const Rows r{p.D, p.cfg.n_pursuers};
  const int env0 = blockIdx.x * kEPB;
  const int nvalid = min(kEPB, p.N - env0);
  if (o.lidar) stream_ones(p.cfg, o.lidar, env0, nvalid);
  stage_block(p, sm, r, env0);
  __syncthreads();
  precompute_block(p.cfg, sm, r);
  __syncthreads();
  if (threadIdx.x < kEPB && (int)threadIdx.x < nvalid) {
    const int lane = threadIdx.x;
    SView v{GView{p.dstate, p.estate, p.D, p.Npad, env0 + lane, r.P}, sm, lane, r, p.D, r.P, env0 + lane, true};
    // right after a reset the Delta=1 snapshot does not exist yet: empty sphere
    if (v.egi(TE_E_STEP) != 0) sm[r.hitmask() * kEPB + lane] = resolve_hits(v, armed_mask(v));
  }
  __syncthreads();
  emit_rows(p.cfg, sm, r, o, env0, nvalid, threadIdx.x, blockDim.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ones must have landed before any patch
  __syncthreads();
  patch_hits(p.cfg, sm, r, o.lidar, nullptr, env0, nvalid);
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
  size_t lds_bytes;
  size_t stack_lds_bytes = 0;  // stacked_kernel (level5)
  int push_split = 0;          // 1 = ring_push_kernel<18, kPushSplit>: the binning of a (chunk, wingman) pair dealt over kPushSplit waves (small shards; TE_PUSH_SPLIT=0/1)
  int stack_regs = 0;          // 18 / 37: ring_push_kernel<DM> + stack_view_kernel<DM> (te_stackview.hpp); 0: stacked_kernel (more than 37 drones, TE_STACKED=lds)
  size_t view_lds_bytes = 0;
  size_t dbg_words = 0;       // diagnostic builds: length of p.dbg
  int engage_regs = 0;         // 1 = engage_kernel<2, 9>, 2 = engage_kernel<6, 12> (level4 family), 3 = engage_stage02_kernel<2, 8>, 4 = engage_stage01_kernel (te_engage.hpp: the env in registers, one wave per
                               // chunk); 0 = engage_observe_kernel (LDS phases): other shapes, stage01 / stage02, TE_ENGAGE=lds
  int k1_help = 0;             // 1 = substeps_kernel<..., HELP>: a noise-helper wave next to every flight (small shards; TE_K1_HELP=0/1)
  int engage_slots = 0;        // 1 = engage_slots_kernel / engage_slots_stage02_kernel (te_engage_slots.hpp: one wave per (chunk, slot)) instead of engage_kernel;
                               // 2 = engage_slots_multi_kernel<slot_spw, own sphere> with slot_waves waves (level5 family, level5_2bt, large level4 shards); TE_ENGAGE=regs turns them off
  int slot_spw = 1, slot_waves = 0; size_t slot_lds = 0;
  int slot_wpe8 = 0;           // engage_slots_kernel<16, 8>: the 64-VGPR build, large shards (TE_SLOT_WPE8=0/1)
  int k2_threads = 256;        // engage/observe kernel: 512 when its LDS allows only two blocks per CU
  float* zero_actions = nullptr;     // te_step_students: the [N,4] action batch nobody reads (every pursuer is scripted)
  uint32_t* ally_scratch = nullptr;  // te_observe_wingman: owner planes between its two launches (te_create allocates them when a wingman is caller-driven)
  // cfg.io_location == TE_IO_HOST: device staging of every I/O buffer of te_step / te_observe / te_reset / te_random_actions /
  // te_get_state / te_set_state (one allocation, carved at 256-byte boundaries by te_create)
  struct HostStage {
    char* base = nullptr;
    float *actions = nullptr, *lidar = nullptr, *inertial = nullptr, *last_action = nullptr, *reward = nullptr;
    float *t_lidar = nullptr, *t_inertial = nullptr, *t_last_action = nullptr;
    uint8_t *done = nullptr, *mask = nullptr; int32_t* info = nullptr; uint32_t* blob = nullptr;
    float* t_gather = nullptr;   // persistent observation under host I/O: where the done envs' terminal LIDAR rows are compacted (the
                                 // observation staging must keep its content then); its own allocation, made by te_set_persistent_obs
  } hs;
  std::vector<int32_t> done_idx;   // host I/O: the done envs of the step, and the host landing zone of their terminal rows
  std::vector<char> done_rows;
  int fill_mode = 1;       // TE_FILL_MODE: 0 pointer loop, 1 scalar buffer loop, 2 + s_setprio 3
  int n_fill_waves = 256;  // fill waves of the sub-step kernel: one per CU of an MI355X; two per CU for the six-sphere background of
                           // level5 (1.6 GB next to ~8 flight waves per SIMD: 128 / 256 / 512 / 1 024 / 2 048 waves -> 599 / 596 / 592 / 619 / 637 us
                           // per step with the scalar loop; stage03 loses 12 / 35 % with 512 / 1 024); TE_FILL_WAVES overrides
  // profiling (te_profile_begin / te_profile_end)
  std::vector<hipEvent_t> events;
  int prof_cap = 0, prof_used = 0;
  // te_set_persistent_obs: cells patched by the previous stacked observation (StackParams::prev), and which buffer they describe
  bool persist_on = false; uint16_t* prev_cells = nullptr; const float* last_stacked = nullptr; int prev_observers = 0, last_n_obs = 0;
  bool prof_markers = false;   // TE_PROF=markers: marker packets between the launches instead of the kernels' own start / stop events
};

static thread_local std::string g_err;
static void launch_census(te_env* e, hipStream_t st);
extern "C" __attribute__((visibility("default"))) void te_destroy(te_env* e);
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

static void launch_census(te_env* e, hipStream_t st) {
  launch_by_family(e->family, [&](auto fam) {
    hipLaunchKernelGGL((census_kernel<decltype(fam)::value>), dim3(e->p.Npad / 64), dim3(64), 0, st, e->p);
  });
}


// ---- cfg.io_location == TE_IO_HOST: the caller's pointers are HOST pointers (what an SB3 SubprocVecEnv-style caller holds).  Inputs
// are copied to the staging buffers, the device entry point runs on them, outputs are copied back and the stream is drained: the
// call returns with the host buffers filled.  The observation (4.1 KB per env) crosses PCIe every step: 12.4 M env-steps/s at 65 536 envs
// with pageable numpy arrays, 13.0 M with pinned ones (52-54 GB/s, tools/host_io_bench.py).
// rows[idx[b]] -> out[b] (row_words floats each): the terminal observations of the done envs, packed for one PCIe copy
__global__ __launch_bounds__(64) void gather_rows_kernel(const float* __restrict__ rows, float* __restrict__ out, const int32_t* __restrict__ idx, int row_words) {
  const size_t from = (size_t)idx[blockIdx.x] * row_words, to = (size_t)blockIdx.x * row_words;
  for (int k = threadIdx.x; k < row_words; k += 64) out[to + k] = rows[from + k];
}
static bool host_io(const te_env* e) { return e && e->p.cfg.io_location == TE_IO_HOST; }
#define TE_H2D(dst, src, bytes) do { if ((src) && (bytes)) TE_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, st)); } while (0)
#define TE_D2H(dst, src, bytes) do { if ((dst) && (bytes)) TE_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, st)); } while (0)

// A launch whose start / stop events carry the dispatch's own begin / end timestamps (hipExtLaunchKernel); the arguments are converted to the
// kernel's formal types first, as a <<<>>> launch would
template <typename... Formals, typename... Actuals>
static void launch_timed(void (*kernel)(Formals...), dim3 grid, dim3 block, unsigned lds, hipStream_t st, hipEvent_t start, hipEvent_t stop, Actuals&&... actuals) {
  static_assert(sizeof...(Formals) == sizeof...(Actuals), "argument count");
  std::tuple<Formals...> held{std::forward<Actuals>(actuals)...};
  void* ptrs[sizeof...(Formals)];
  std::apply([&](auto&... a) { int i = 0; ((ptrs[i++] = (void*)&a), ...); }, held);
  (void)hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, st, start, stop, 0);
}

extern "C" {

__attribute__((visibility("default"))) const char* te_last_error(void) { return g_err.c_str(); }

__attribute__((visibility("default"))) int te_create(const te_config* cfg, int32_t device_id, te_env** out) {
  if (!cfg || !out) return fail("te_create: null argument");
  if (cfg->struct_size != sizeof(te_config)) return fail("te_create: te_config.struct_size mismatch (ABI skew)");
  const int D = cfg->n_pursuers + cfg->n_invaders;
  if (cfg->n_envs < 1) return fail("te_create: n_envs < 1");
  if (cfg->n_pursuers < 1 || cfg->n_invaders < 1 || D > kMaxD64 || cfg->n_pursuers > 31) return fail("te_create: need 1 <= P <= 31, 1 <= I, P + I <= 64");
  if (cfg->initial_invaders < 0 || cfg->invaders_per_round < 0) return fail("te_create: initial_invaders / invaders_per_round must not be negative");
  if (cfg->substeps < 0 || cfg->substeps > 255) return fail("te_create: substeps out of range");
  // substeps == 0: env.step() without physics (the IMU is read from the state as it is, then engagement / reward / termination /
  // waves / observation as usual).  Used to replay the reference's task-logic fixtures on exactly their positions.
  if (cfg->substeps == 0 && cfg->observe_lag != 0) return fail("te_create: substeps == 0 (no physics) needs observe_lag == 0");
  if (cfg->task < TE_TASK_STAGE01 || cfg->task > TE_TASK_LEVEL5_FUSION) return fail("te_create: unknown task");
  if (cfg->ground_contact && family_of(cfg->task) != FAM_LEVEL4) return fail("te_create: cfg.ground_contact is built for the level4 task family only");
  if (cfg->evaluation && (((uint32_t)cfg->evaluation >> 8) >> cfg->n_pursuers) != 0u) return fail("te_create: cfg.evaluation's driver mask names a pursuer that does not exist");
  if (cfg->evaluation && !(family_of(cfg->task) == FAM_LEVEL4 && cfg->ally_policy == TE_ALLY_BT && !cfg->stacked_obs))
    return fail("te_create: cfg.evaluation (Evaluation_Task rules) is the level4 task family with behaviour-tree drivers and the own-sphere observation");
  if (cfg->ally_policy == TE_ALLY_EXTERNAL && !(family_of(cfg->task) == FAM_LEVEL4 && cfg->n_pursuers == 2))
    return fail("te_create: TE_ALLY_EXTERNAL (exp05) is the level4 task family with exactly 2 pursuers (exp05_vFinal_task.py:103)");
  if (cfg->stacked_obs && family_of(cfg->task) != FAM_LEVEL4) return fail("te_create: stacked_obs needs a level4-family task");
  if (cfg->task == TE_TASK_STAGE01 && !(cfg->n_pursuers == 2 && cfg->n_invaders == 1)) return fail("te_create: stage01 is 2 pursuers + 1 invader");
  if (cfg->lidar_radius <= 0.0f || cfg->dome_radius <= 0.0f || cfg->max_speed <= 0.0f) return fail("te_create: radii / max_speed must be positive");
  if (cfg->control_every_substep != 0 && cfg->control_every_substep != 1) return fail("te_create: control_every_substep is 0 or 1");
  if (cfg->lidar_channels != 3 && cfg->lidar_channels != 2) return fail("te_create: lidar_channels is 3 (distance, flag, time) or 2 (distance, flag)");
  if (cfg->lidar_channels == 2 && cfg->stacked_obs) return fail("te_create: the stacked observation (level5) always has 3 channels");
  if (cfg->io_location != TE_IO_DEVICE && cfg->io_location != TE_IO_HOST) return fail("te_create: io_location is TE_IO_DEVICE or TE_IO_HOST");
  if (cfg->io_location == TE_IO_HOST && (cfg->stacked_obs || cfg->ally_policy == TE_ALLY_EXTERNAL || ((uint32_t)cfg->evaluation >> 8) != 0u))
    return fail("te_create: TE_IO_HOST serves te_reset / te_observe / te_step / te_random_actions / te_get_state / te_set_state; the stacked observation and caller-driven wingmen take device pointers");
  if (cfg->drone_contact != 0 && cfg->drone_contact != 1) return fail("te_create: drone_contact is 0 or 1");
  if (cfg->drone_contact && !(cfg->contact_radius > 0.0f)) return fail("te_create: contact_radius must be positive");
  int ndev = 0;
  TE_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail("te_create: no HIP device visible; this library has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return fail("te_create: bad device_id");
  DeviceGuard guard(device_id);
  if (!guard.ok) return fail("te_create: hipSetDevice failed");
  te_env* e = new (std::nothrow) te_env();
  if (!e) return fail("te_create: out of host memory");
  e->device = device_id;
  e->p.dstate = nullptr; e->p.estate = nullptr; e->p.slot_mask = nullptr; e->p.live_mask = nullptr; e->p.mixed_count = nullptr; e->p.mixed_items = nullptr; e->p.stage_tab = nullptr; e->p.snap = nullptr; e->p.ring = nullptr; e->p.dbg = nullptr;
  // ONE failure path from here on: whatever has been allocated so far is released (te_destroy) before the error is returned
  auto bail = [&](const std::string& why) { te_destroy(e); return fail(why); };
#define TE_HIP_OR_BAIL(x)                                                                              \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) return bail(std::string("te_create: " #x ": ") + hipGetErrorString(e_));      \
  } while (0)
  e->family = family_of(cfg->task);
  e->p.cfg = *cfg;
  e->p.kd = derive(*cfg);
  e->p.N = cfg->n_envs; e->p.D = D; e->p.Npad = (cfg->n_envs + 63) / 64 * 64;
  e->lds_bytes = (size_t)lds_rows(D, cfg->n_pursuers) * kEPB * sizeof(uint32_t);
  if (e->family == FAM_LEVEL4) {
    if (cfg->n_pursuers <= 2 && D <= 11) e->engage_regs = 1;
    else if (cfg->n_pursuers <= 6 && D <= 18) e->engage_regs = 2;
    else if (cfg->n_pursuers <= 7 && D <= 37) e->engage_regs = 5;
  } else if (e->family == FAM_STAGE02) {
    if (cfg->n_pursuers <= 2 && D <= 10) e->engage_regs = 3;
  } else if (e->family == FAM_STAGE01) e->engage_regs = 4;
  if (const char* v = getenv("TE_ENGAGE")) { if (!strcmp(v, "lds")) e->engage_regs = 0; }
  const bool regs_l4 = e->engage_regs == 1 || e->engage_regs == 2 || e->engage_regs == 5;
  // one wave per (chunk, slot) instead of one wave per chunk: a shard with fewer chunks than the chip has SIMDs is one dependent chain per
  // wave, and the chain is what the slot waves shorten (te_engage_slots.hpp); large shards keep engage_kernel (fewer instructions in total)
  if (regs_l4 && !cfg->stacked_obs && D <= kSlotWaves && cfg->n_pursuers <= kSlotPursuers && cfg->reward_model == TE_REWARD_EXP03 && !cfg->drone_contact) {
    e->engage_slots = (long long)cfg->n_envs * D <= kSlotsMaxPairs ? 1 : 0;
    if (const char* v = getenv("TE_ENGAGE")) { if (!strcmp(v, "slots")) e->engage_slots = 1; else if (!strcmp(v, "regs")) e->engage_slots = 0; }
  }
  // several slots per wave (engage_slots_multi_kernel): the level5 family (no own sphere; more drones than a workgroup has waves) and the level4
  // family beyond 16 drones (level5_2bt: 2 + 30).  For the level5 family a large shard gets fewer, fatter waves: when the chip cannot hold every
  // chunk's workgroup at once (256 CUs x 24 waves at <= 80 VGPRs) — level5 x 65 536 envs: nine-wave workgroups ran in two rounds, 35 us; six
  // waves of three slots: 28 us
  if (regs_l4 && (D <= 32 || (cfg->stacked_obs && D <= 3 * kSlotWaves)) && !cfg->drone_contact && (cfg->stacked_obs || D > kSlotWaves || e->engage_slots == 1)) {
    int spw = D <= kSlotWaves ? 1 : D <= 2 * kSlotWaves ? 2 : 3;   // (beyond 32 drones — level5_fusion 36, level5_dumb 37 — the masks are 64 bits: <3, false, WIDE>)
    const long long chunks = (cfg->n_envs + 63) / 64;
    const int spw_max = cfg->stacked_obs ? 3 : 2;
    // (with the own sphere, on shapes one slot per wave serves too, engage_slots_kernel stays ahead at every size — stage03 x 65 536: 81.8 vs
    // 86.2 us per step, x 32 768: 51.0 vs 53.1, profiles/r04_q_ab_slots_per_wave.txt: most of its eleven waves retire at once — TE_SLOT_SPW=2 forces the other)
    while (cfg->stacked_obs && spw < spw_max && chunks * ((D + spw - 1) / spw) > 256 * 24 && cfg->n_pursuers <= (D + spw) / (spw + 1)) spw += 1;
    if (const char* v = getenv("TE_SLOT_SPW")) { const int k = atoi(v); if (k >= 1 && k <= spw_max && (D + k - 1) / k <= kSlotWaves) spw = k; }
    const int W = (D + spw - 1) / spw;
    const size_t lds = (size_t)multi_slot_lds_rows(D, cfg->n_pursuers, !cfg->stacked_obs) * 256;
    bool ok = cfg->n_pursuers <= W && lds <= 160 * 1024 && (cfg->stacked_obs || spw > 1);   // (one slot per wave with the own sphere: engage_slots_kernel)
    if (const char* v = getenv("TE_ENGAGE")) { if (!strcmp(v, "regs")) ok = false; }
    if (ok && lds > 64 * 1024) {   // beyond the default dynamic-LDS limit: level5_2bt's 32 drones with {cell, range} rows
      const void* fn = cfg->stacked_obs ? (D > 32 ? (const void*)engage_slots_multi_kernel<3, false, true> : spw == 3 ? (const void*)engage_slots_multi_kernel<3, false> : spw == 2 ? (const void*)engage_slots_multi_kernel<2, false> : (const void*)engage_slots_multi_kernel<1, false>)
                                        : (const void*)engage_slots_multi_kernel<2, true>;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); ok = false; }
    }
    if (ok) { e->engage_slots = 2; e->slot_spw = spw; e->slot_waves = W; e->slot_lds = lds; }
  }
  // (opt-in: at 65 536 envs the 64-VGPR build is ahead once the rollout is heavy — steady state 679 -> 700 M env-steps/s — and behind in the
  // light window right after a reset, which is the one the driver's bench times: 910 -> 890 M; profiles/r04_w_ab_slot_waves_64_vgprs.txt)
  e->slot_wpe8 = 0;
  if (const char* v = getenv("TE_SLOT_WPE8")) e->slot_wpe8 = e->engage_slots == 1 && e->family == FAM_LEVEL4 && atoi(v) != 0;
  if (e->family == FAM_STAGE02 && e->engage_regs == 3) {   // stage02 in the same form (engage_slots_stage02_kernel)
    e->engage_slots = (long long)cfg->n_envs * D <= kSlotsMaxPairs ? 1 : 0;
    if (const char* v = getenv("TE_ENGAGE")) { if (!strcmp(v, "slots")) e->engage_slots = 1; else if (!strcmp(v, "regs")) e->engage_slots = 0; }
  }
  if (D > kMaxD && !(regs_l4 && cfg->stacked_obs))
    return bail("te_create: more than 32 drones per env are served for the stacked-observation tasks only (te_step_stacked / te_step_students: stacked_obs, P <= 7, P + I <= 37)");
  if ((cfg->agent_scripted || cfg->reward_model != TE_REWARD_EXP03 || !cfg->agent_death_terminates || cfg->initial_invaders != 1 || cfg->invaders_per_round != 1) && !regs_l4)
    return bail("te_create: agent_scripted / reward_model / agent_death_terminates / the round rule are built into engage_kernel: the level4 task family with P <= 7 and P + I <= 37");
  if (cfg->drone_contact && !(e->engage_regs == 1 || e->engage_regs == 2))
    return bail("te_create: cfg.drone_contact is built into engage_kernel: the level4 task family with P <= 6 and P + I <= 18");
  e->p.dense_min = kDenseMin;
  // noise-helper waves (substeps_kernel<..., HELP>): worth their issue slots only while SIMDs idle, i.e. on small shards
  const bool help_ok = cfg->motor_noise && cfg->substeps >= 2 && cfg->substeps <= 48 && cfg->n_pursuers <= kEagerP;   // (the helped flights request eagerly: fly<..., EAGER>)
  // ... or, in the level4 family, where a rollout flies the pursuers and one or two invaders of the D slots, shards whose TYPICAL launch stays below
  // ~1.25 flights per SIMD: with the eager requests the helped kernel is ahead up to 20 480 stage03 envs in the window after a reset (+ 5 %) and in
  // the steady state (+ 7 %), and 6 % behind with every slot armed; a tie at 24 576, 8 % behind at 32 768 (profiles/r04_k_ab_help_mid_shards.txt)
  const long long chunks_ = (cfg->n_envs + 63) / 64;
  const bool typical_light = e->family == FAM_LEVEL4 && chunks_ * std::min(D, cfg->n_pursuers + 2) <= 1280;
  e->k1_help = (help_ok && ((long long)cfg->n_envs * D <= kHelpMaxPairs || typical_light)) ? 1 : 0;
  if (const char* v = getenv("TE_K1_HELP")) e->k1_help = (atoi(v) != 0 && help_ok) ? 1 : 0;
  if (const char* v = getenv("TE_DENSE_MIN")) { int n = atoi(v); if (n >= 1 && n <= 65) e->p.dense_min = n; }
  if (cfg->stacked_obs) e->n_fill_waves = 512;
  if (const char* v = getenv("TE_FILL_MODE")) e->fill_mode = atoi(v);
  if (const char* v = getenv("TE_FILL_WAVES")) { int n = atoi(v); if (n >= 1 && n <= (1 << 20)) e->n_fill_waves = n; }
  {
    hipError_t le = hipSuccess;
    launch_by_family(e->family, [&](auto fam) {
      le = hipFuncSetAttribute(reinterpret_cast<const void*>(&engage_observe_kernel<decltype(fam)::value>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
    });
    // > 53 KB of LDS per block = at most two blocks per CU: run those with eight waves (level4 family only: D > 15)
    e->k2_threads = (e->family == FAM_LEVEL4 && e->lds_bytes > 53 * 1024) ? 512 : 256;
    if (const char* v = getenv("TE_K2_THREADS")) { int n = atoi(v); if ((n == 256 || n == 512) && e->family == FAM_LEVEL4) e->k2_threads = n; }
    if (le == hipSuccess && e->k2_threads == 512)
      le = hipFuncSetAttribute(reinterpret_cast<const void*>(&engage_observe_kernel<FAM_LEVEL4, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
    if (le == hipSuccess)
      le = hipFuncSetAttribute(reinterpret_cast<const void*>(&observe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
    if (le != hipSuccess) return bail(std::string("te_create: this many drones per env needs more LDS than a workgroup may have: ") + hipGetErrorString(le));
  }
  const size_t dwords = (size_t)(TE_DRONE_WORDS + TE_X_WORDS) * D * e->p.Npad, ewords = (size_t)TE_ENV_WORDS * e->p.Npad;
  if (dwords >= (1ull << 30)) return bail("te_create: n_envs * drones too large for one te_env (state planes are indexed with 32 bits); shard it");
  if (hipMalloc(&e->p.dstate, dwords * 4) != hipSuccess || hipMalloc(&e->p.estate, ewords * 4) != hipSuccess)
    return bail("te_create: hipMalloc failed");
  if (hipMalloc(&e->p.slot_mask, (size_t)(e->p.Npad / 64) * 8) != hipSuccess || hipMalloc(&e->p.live_mask, (size_t)(e->p.Npad / 64) * 8) != hipSuccess || hipMalloc(&e->p.mixed_count, (size_t)(e->p.Npad / 64) * 4) != hipSuccess ||
      hipMalloc(&e->p.mixed_items, (size_t)(e->p.Npad / 64) * kMixedCap * sizeof(uint16_t)) != hipSuccess)
    return bail("te_create: hipMalloc failed");
  TE_HIP_OR_BAIL(hipMemsetAsync(e->p.mixed_count, 0, (size_t)(e->p.Npad / 64) * 4, nullptr));
  {
    const Rows r{D, cfg->n_pursuers};
    std::vector<uint32_t> tab((size_t)r.staged());
    for (int row = 0; row < r.staged(); ++row) tab[(size_t)row] = staged_row_offset(e->p, r, row);
    if (hipMalloc(&e->p.stage_tab, tab.size() * 4) != hipSuccess ||
        hipMemcpy(e->p.stage_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return bail("te_create: hipMalloc failed");
  }
  // caller-driven wingmen (exp05's ally, Evaluation_Task's "nn" drivers): te_observe_wingman's scratch planes are allocated
  // HERE, not on first use: the entry point only enqueues work (it must be legal under stream capture and never synchronise)
  if (cfg->ally_policy == TE_ALLY_EXTERNAL || ((uint32_t)cfg->evaluation >> 8) != 0u) {
    if (hipMalloc(&e->ally_scratch, (size_t)2 * D * e->p.Npad * 4) != hipSuccess) return bail("te_create: hipMalloc failed");
  }
  if (cfg->io_location == TE_IO_HOST) {
    const size_t N = (size_t)cfg->n_envs, lw = (size_t)lidar_words(*cfg) * 4;
    const size_t blob = (N * ((size_t)D * TE_DRONE_WORDS + TE_ENV_WORDS)) * 4;
    const size_t sizes[12] = {N * 16, N * lw, N * 60, N * 16, N * 4, N * lw, N * 60, N * 16, N, N, N * 16, blob};
    size_t off[13]; off[0] = 0;
    for (int k = 0; k < 12; ++k) off[k + 1] = off[k] + ((sizes[k] + 255) & ~(size_t)255);
    if (hipMalloc(&e->hs.base, off[12]) != hipSuccess) return bail("te_create: hipMalloc failed (host I/O staging)");
    char* b = e->hs.base;
    e->hs.actions = (float*)(b + off[0]); e->hs.lidar = (float*)(b + off[1]); e->hs.inertial = (float*)(b + off[2]);
    e->hs.last_action = (float*)(b + off[3]); e->hs.reward = (float*)(b + off[4]); e->hs.t_lidar = (float*)(b + off[5]);
    e->hs.t_inertial = (float*)(b + off[6]); e->hs.t_last_action = (float*)(b + off[7]); e->hs.done = (uint8_t*)(b + off[8]);
    e->hs.mask = (uint8_t*)(b + off[9]); e->hs.info = (int32_t*)(b + off[10]); e->hs.blob = (uint32_t*)(b + off[11]);
  }
  e->p.entry_words = TE_RING_ENTRY_WORDS(D);
  if (cfg->stacked_obs && (cfg->agent_scripted || cfg->evaluation)) {
    if (hipMalloc(&e->zero_actions, (size_t)e->p.Npad * 16) != hipSuccess) return bail("te_create: hipMalloc failed");
    TE_HIP_OR_BAIL(hipMemsetAsync(e->zero_actions, 0, (size_t)e->p.Npad * 16, nullptr));
  }
  if (cfg->stacked_obs) {
    e->stack_lds_bytes = (size_t)stack_lds_rows(D, cfg->n_pursuers) * kEPB * sizeof(uint32_t);
    const size_t snap_bytes = (size_t)snap_words(D, cfg->n_pursuers) * e->p.Npad * 4;
    const size_t ring_bytes = (size_t)cfg->n_envs * cfg->n_pursuers * TE_RING_DEPTH * e->p.entry_words * 4;
    hipError_t le = hipFuncSetAttribute(reinterpret_cast<const void*>(&stacked_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->stack_lds_bytes);
    e->stack_regs = D <= 18 ? 18 : (D <= 37 ? 37 : 0);
    if (const char* v = getenv("TE_STACKED")) { if (!strcmp(v, "lds")) e->stack_regs = 0; }
    // small shards: fewer push waves than SIMDs even after dealing each pair's binning over kPushSplit waves
    e->push_split = e->stack_regs == 18 && (long long)((cfg->n_envs + 63) / 64) * cfg->n_pursuers * kPushSplit <= 4096 ? 1 : 0;
    if (const char* v = getenv("TE_PUSH_SPLIT")) e->push_split = e->stack_regs == 18 && atoi(v) != 0;
    e->view_lds_bytes = (size_t)view_lds_rows(D) * kEPB * sizeof(uint32_t);
    if (le == hipSuccess && e->stack_regs == 18) le = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_view_kernel<18>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->view_lds_bytes);
    if (le == hipSuccess && e->stack_regs == 37) le = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_view_kernel<37>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->view_lds_bytes);
    if (le != hipSuccess || hipMalloc(&e->p.snap, snap_bytes) != hipSuccess || hipMalloc(&e->p.ring, ring_bytes) != hipSuccess)
      return bail("te_create: stacked observation buffers (ring / snapshot / LDS) could not be set up");
    TE_HIP_OR_BAIL(hipMemsetAsync(e->p.snap, 0, snap_bytes, nullptr));
    TE_HIP_OR_BAIL(hipMemsetAsync(e->p.ring, 0, ring_bytes, nullptr));
  }
#ifdef TE_DEBUG_STAMPS
  const size_t dbg_words = 64 + 16 * (size_t)(e->p.Npad / kEPB + 1) + 4 * ((size_t)(e->p.D + kMixedWaves) * (e->p.Npad >> 6) + 4096);  // engage records, then the sub-step kernel's
  e->dbg_words = dbg_words;
  if (hipMalloc(&e->p.dbg, dbg_words * sizeof(unsigned long long)) != hipSuccess) e->p.dbg = nullptr;
  else { (void)hipMemset(e->p.dbg, 0, dbg_words * sizeof(unsigned long long)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_te_dbg), &e->p.dbg, sizeof(e->p.dbg)); }
#endif
  TE_HIP_OR_BAIL(hipMemsetAsync(e->p.dstate, 0, dwords * 4, nullptr));
  TE_HIP_OR_BAIL(hipMemsetAsync(e->p.estate, 0, ewords * 4, nullptr));
  hipLaunchKernelGGL(init_planes, dim3(256), dim3(256), 0, nullptr, e->p);
  const int blocks = (e->p.N + 255) / 256;
  launch_by_family(e->family, [&](auto fam) {
    hipLaunchKernelGGL((reset_kernel<decltype(fam)::value>), dim3(blocks), dim3(256), 0, nullptr, e->p, (const uint8_t*)nullptr);
  });
  launch_census(e, nullptr);
  TE_HIP_OR_BAIL(hipGetLastError());
  TE_HIP_OR_BAIL(hipStreamSynchronize(nullptr));
#undef TE_HIP_OR_BAIL
  *out = e;
  return 0;
}

__attribute__((visibility("default"))) void te_destroy(te_env* e) {
  if (!e) return;
  DeviceGuard guard(e->device);
  for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
  (void)hipFree(e->p.dstate);
  (void)hipFree(e->p.estate);
  (void)hipFree(e->ally_scratch);
  (void)hipFree(e->zero_actions);
  (void)hipFree(e->hs.base); (void)hipFree(e->hs.t_gather);
  (void)hipFree(e->p.slot_mask); (void)hipFree(e->p.live_mask); (void)hipFree(e->p.mixed_count); (void)hipFree(e->p.mixed_items);
  (void)hipFree(e->p.stage_tab);
  if (e->p.snap) (void)hipFree(e->p.snap);
  if (e->p.ring) (void)hipFree(e->p.ring);
  if (e->prev_cells) (void)hipFree(e->prev_cells);
  if (e->p.dbg) (void)hipFree(e->p.dbg);
  delete e;
}

__attribute__((visibility("default"))) int te_reset(te_env* e, const uint8_t* env_mask, void* stream) {
  if (!e) return fail("te_reset: null env");
  DeviceGuard guard(e->device);
  if (host_io(e) && env_mask) {   // the mask is a host array: staged
    hipStream_t st = (hipStream_t)stream;
    TE_H2D(e->hs.mask, env_mask, (size_t)e->p.N);
    TE_HIP(hipStreamSynchronize(st));   // the caller may reuse or free its (possibly pinned) array as soon as the call returns
    env_mask = e->hs.mask;
  }
  const int blocks = (e->p.N + 255) / 256;
  launch_by_family(e->family, [&](auto fam) {
    hipLaunchKernelGGL((reset_kernel<decltype(fam)::value>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, e->p, env_mask);
  });
  launch_census(e, (hipStream_t)stream);
  TE_HIP(hipGetLastError());
  return 0;
}

// Evaluation_Task.compute_info (evaluation_task.py:553-574): (lw_kills, lw_alive, lw_munitions, current_wave, step) per pursuer
__global__ __launch_bounds__(256) void wingman_info_kernel(Params p, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x, P = p.cfg.n_pursuers;
  if (i >= p.N * P) return;
  const int env = i / P, s = i - env * P;
  const GView v{p.dstate, p.estate, p.D, p.Npad, env, P};
  int32_t* row = out + (size_t)i * 5;
  row[0] = v.gi(TE_D_KILLS, s); row[1] = v.gi(TE_D_ARMED, s) ? 1 : 0; row[2] = v.gi(TE_D_MUNITION, s);
  row[3] = v.egi(TE_E_INFO_WAVE); row[4] = v.egi(TE_E_STEP);
}
__attribute__((visibility("default"))) int te_wingman_info(te_env* e, int32_t* wingman_info, void* stream) {
  if (!e || !wingman_info) return fail("te_wingman_info: null argument");
  if (!e->p.cfg.evaluation) return fail("te_wingman_info: per-wingman kills are only counted under cfg.evaluation (Evaluation_Task)");
  DeviceGuard guard(e->device);
  const int n = e->p.N * e->p.cfg.n_pursuers;
  hipLaunchKernelGGL(wingman_info_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->p, wingman_info);
  TE_HIP(hipGetLastError());
  return 0;
}

static bool wingman_is_callers(const te_env* e, int wingman) {
  const te_config& c = e->p.cfg;
  if (wingman < 0 || wingman >= c.n_pursuers) return false;
  return (c.ally_policy == TE_ALLY_EXTERNAL && wingman == 1) || (((uint32_t)c.evaluation >> (8 + wingman)) & 1u) != 0u;
}

__attribute__((visibility("default"))) int te_observe_wingman(te_env* e, int32_t wingman, float* ally_lidar, float* ally_inertial,
                                                              float* ally_last_action, uint8_t* ally_active, void* stream) {
  if (!e) return fail("te_observe_wingman: null env");
  if (!wingman_is_callers(e, wingman))
    return fail("te_observe_wingman: this pursuer is not driven by the caller (exp05: cfg.ally_policy == TE_ALLY_EXTERNAL, pursuer 1; evaluation: the driver mask in cfg.evaluation)");
  if ((ally_lidar && ((uintptr_t)ally_lidar & 15)) || (ally_last_action && ((uintptr_t)ally_last_action & 15)))
    return fail("te_observe_wingman: lidar and last_action must be 16-byte aligned");
  DeviceGuard guard(e->device);
  hipStream_t st = (hipStream_t)stream;
  const size_t n_floats = (size_t)e->p.N * lidar_words(e->p.cfg);
  const int nchunks = e->p.Npad / kEPB;
  if (ally_lidar && (n_floats & 3) == 0 && (n_floats >> 2) < (1ull << 32) && e->p.N >= 4096) {  // full-size batches: two launches
    if (!e->ally_scratch) return fail("te_observe_wingman: internal error: no scratch planes (te_create allocates them for caller-driven wingmen)");
    hipLaunchKernelGGL(ally_view_kernel, dim3(256 + nchunks), dim3(64), 0, st, e->p, (int)wingman, ally_lidar, (uint32_t)(n_floats >> 2), 256u,
                       ally_inertial, ally_last_action, ally_active, e->ally_scratch);
    hipLaunchKernelGGL(ally_patch_kernel, dim3((unsigned)(((size_t)e->p.D * e->p.Npad + 255) / 256)), dim3(256), 0, st, e->p, ally_lidar, e->ally_scratch);
  } else {
    hipLaunchKernelGGL(observe_ally_kernel, dim3(nchunks), dim3(256), 0, st, e->p, (int)wingman, ally_lidar, ally_inertial, ally_last_action, ally_active);
  }
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_set_wingman_actions(te_env* e, int32_t wingman, const float* ally_actions, void* stream) {
  if (!e) return fail("te_set_wingman_actions: null env");
  if (!wingman_is_callers(e, wingman))
    return fail("te_set_wingman_actions: this pursuer is not driven by the caller (exp05: cfg.ally_policy == TE_ALLY_EXTERNAL, pursuer 1; evaluation: the driver mask in cfg.evaluation)");
  if (!ally_actions || ((uintptr_t)ally_actions & 15)) return fail("te_set_wingman_actions: actions ([N,4] f32, 16-byte aligned) is required");
  DeviceGuard guard(e->device);
  hipLaunchKernelGGL(set_ally_actions_kernel, dim3((e->p.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->p, (int)wingman, ally_actions);
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_observe_ally(te_env* e, float* ally_lidar, float* ally_inertial, float* ally_last_action,
                                                           uint8_t* ally_active, void* stream) {
  if (!e) return fail("te_observe_ally: null env");
  if (e->p.cfg.ally_policy != TE_ALLY_EXTERNAL) return fail("te_observe_ally: this te_env's ally is not driven by the caller (cfg.ally_policy != TE_ALLY_EXTERNAL)");
  return te_observe_wingman(e, 1, ally_lidar, ally_inertial, ally_last_action, ally_active, stream);
}

__attribute__((visibility("default"))) int te_set_ally_actions(te_env* e, const float* ally_actions, void* stream) {
  if (!e) return fail("te_set_ally_actions: null env");
  if (e->p.cfg.ally_policy != TE_ALLY_EXTERNAL) return fail("te_set_ally_actions: this te_env's ally is not driven by the caller (cfg.ally_policy != TE_ALLY_EXTERNAL)");
  if (!ally_actions || ((uintptr_t)ally_actions & 15)) return fail("te_set_ally_actions: ally_actions ([N,4] f32, 16-byte aligned) is required");
  return te_set_wingman_actions(e, 1, ally_actions, stream);
}

static int observe_device(te_env* e, float* obs_lidar, float* obs_inertial, float* obs_last_action, void* stream);
__attribute__((visibility("default"))) int te_observe(te_env* e, float* obs_lidar, float* obs_inertial, float* obs_last_action, void* stream) {
  if (!host_io(e)) return observe_device(e, obs_lidar, obs_inertial, obs_last_action, stream);
  DeviceGuard guard(e->device);
  hipStream_t st = (hipStream_t)stream;
  const size_t N = (size_t)e->p.N;
  if (observe_device(e, obs_lidar ? e->hs.lidar : nullptr, obs_inertial ? e->hs.inertial : nullptr, obs_last_action ? e->hs.last_action : nullptr, stream)) return 1;
  TE_D2H(obs_lidar, e->hs.lidar, N * lidar_words(e->p.cfg) * 4); TE_D2H(obs_inertial, e->hs.inertial, N * 60); TE_D2H(obs_last_action, e->hs.last_action, N * 16);
  TE_HIP(hipStreamSynchronize(st));
  return 0;
}
static int observe_device(te_env* e, float* obs_lidar, float* obs_inertial, float* obs_last_action, void* stream) {
  if (!e) return fail("te_observe: null env");
  if (e->p.D > kMaxD) return fail("te_observe: more than 32 drones per env: the observation comes out of te_observe_stacked / te_step_stacked (or te_step_students)");
  if ((obs_lidar && ((uintptr_t)obs_lidar & 15)) || (obs_last_action && ((uintptr_t)obs_last_action & 15)))
    return fail("te_observe: obs_lidar and obs_last_action must be 16-byte aligned");
  DeviceGuard guard(e->device);
  const int blocks = (e->p.N + kEPB - 1) / kEPB;
  if (obs_lidar && obs_lidar == e->last_stacked) e->last_stacked = nullptr;   // persistent observation: this call rewrites the buffer without recording
  hipLaunchKernelGGL(observe_kernel, dim3(blocks), dim3(256), e->lds_bytes, (hipStream_t)stream, e->p, ObsOut{obs_lidar, obs_inertial, obs_last_action});
  TE_HIP(hipGetLastError());
  return 0;
}

// one env.step of every env.  `obs_lidar` is the buffer whose background the sub-step kernel's fill waves stream:
// [N,3,13,26] for te_step, [N,6,3,13,26] (`lidar_words_per_env` = 6084) for te_step_stacked, which also passes `stack`.
static int step_impl(te_env* e, const float* actions, float* obs_lidar, size_t lidar_words_per_env, float* obs_inertial,
                     float* obs_last_action, float* reward, uint8_t* done, int32_t* info, float* terminal_lidar,
                     float* terminal_inertial, float* terminal_last_action, const StackOut* stack, void* stream, int n_obs = 1) {
  if (!e) return fail("te_step: null env");
  if (!actions || !reward || !done || !info) return fail("te_step: actions, reward, done and info are required");
  if ((e->p.ring != nullptr) != (stack != nullptr))
    return fail(stack ? "te_step_stacked: this te_env was created without cfg.stacked_obs"
                      : "te_step: this te_env keeps a snapshot ring (cfg.stacked_obs); step it with te_step_stacked");
  if (((uintptr_t)actions & 15) || ((uintptr_t)info & 15) || (obs_lidar && ((uintptr_t)obs_lidar & 15)) ||
      (obs_last_action && ((uintptr_t)obs_last_action & 15)) || (terminal_last_action && ((uintptr_t)terminal_last_action & 15)))
    return fail("te_step: actions, info, obs_lidar and the last_action buffers must be 16-byte aligned");
  DeviceGuard guard(e->device);
  hipStream_t st = (hipStream_t)stream;
  const Params& p = e->p;
  // Profiling (te_profile_begin): 4 events per step.  Default: the kernels are launched with hipExtLaunchKernel, whose start / stop events
  // carry the dispatch's own begin / end timestamps (what rocprofv3 --kernel-trace reports): [0, 1] = the sub-step kernel, [2] = start of the
  // engage kernel, [3] = end of the step's last kernel.  TE_PROF=markers: round-2 form, marker packets between the launches (each bracket
  // then includes ~5 us of queue time: tools/event_overhead_probe.py).
  const bool prof = e->prof_used + 4 <= e->prof_cap;
  const bool prof_ext = prof && !e->prof_markers;
  hipEvent_t* pev = prof ? &e->events[e->prof_used] : nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;   // start / stop events of the next TE_LAUNCH
#define TE_LAUNCH(K, grid, block, lds, ...) do { \
    if (ev_a || ev_b) launch_timed(K, grid, block, lds, st, ev_a, ev_b, __VA_ARGS__); \
    else hipLaunchKernelGGL(K, grid, block, lds, st, __VA_ARGS__); } while (0)
  if (prof && !prof_ext) TE_HIP(hipEventRecord(pev[0], st));
  const int waves = p.D * (p.Npad >> 6);
  const bool noise = p.cfg.motor_noise != 0;
  // LIDAR background: N*1014 floats = quads float4 (+ <4 tail floats).  The drone waves write one float4 per
  // lane; the rest is split evenly over the fill waves.
  FillJob fill{nullptr, 0u, 0u, 0u, nullptr, 0u, 0u, 0u, 0u, 0u};
  const size_t n_floats = (size_t)p.N * lidar_words_per_env;
  // persistent observation (te_set_persistent_obs): when this is the buffer the previous stacked observation went to, it still holds
  // ones + the recorded cells, and only those are touched (no background stream); otherwise fill as usual and record
  int persist = 0;
  if (stack && e->persist_on && e->stack_regs && n_obs <= e->prev_observers) persist = (obs_lidar && obs_lidar == e->last_stacked && n_obs == e->last_n_obs) ? (n_obs > 1 ? 3 : 1) : 2;
  if (!stack && e->persist_on && e->engage_regs && obs_lidar) persist = obs_lidar == e->last_stacked ? 1 : 2;   // te_step: the agent's own sphere
  if (obs_lidar && persist == 1) {   // the cells of the previous observation go back to one next to this launch's flights
    const uint32_t lists = stack ? (uint32_t)n_obs * TE_STACK_SPHERES : 1u;
    fill = FillJob{obs_lidar, 0u, lists * (uint32_t)(p.Npad >> 6), 3u, e->prev_cells, lists, (uint32_t)p.D, (uint32_t)p.Npad, (uint32_t)p.N, (uint32_t)(lidar_words_per_env / (size_t)lists)};
  } else if (obs_lidar && persist != 3) {
    const size_t quads = n_floats >> 2;
    if (quads >= 4096 && quads < (1ull << 32)) {
      fill = FillJob{obs_lidar, (uint32_t)quads, (uint32_t)e->n_fill_waves, (quads >> 6) < (1u << 22) ? (uint32_t)e->fill_mode : 0u, nullptr, 0u, 0u, 0u, 0u, 0u};  // the SGPR block offset is 32-bit: < 4 GB
      if (n_floats & 3) hipLaunchKernelGGL(fill_ones_kernel, dim3(1), dim3(64), 0, st, obs_lidar + (quads << 2), n_floats & 3);
    } else {  // tiny or huge buffers: plain fill kernel first
      hipLaunchKernelGGL(fill_ones_kernel, dim3(2048), dim3(256), 0, st, obs_lidar, n_floats);
    }
  }
  if (prof_ext) { ev_a = pev[0]; ev_b = pev[1]; }
  const int b1 = (int)fill.n_fill_waves + waves + (e->family == FAM_LEVEL4 && p.dense_min > 1 ? kMixedWaves * (p.Npad >> 6) : 0);  // + mixed-wave candidates
  launch_by_family(e->family, [&](auto fam) {
    constexpr int F = decltype(fam)::value;
    auto go = [&](auto noise_c, auto fill_c) {
      constexpr bool kNoise = decltype(noise_c)::value;
      if (kNoise && e->k1_help) {   // small shards: every flight (and candidate) block carries a second wave that prepares the motor noise
        const size_t lds = (size_t)p.cfg.substeps * 1024 + 16;
        if (p.cfg.control_every_substep)
          TE_LAUNCH((substeps_kernel<F, kNoise, decltype(fill_c)::value, true, kNoise>), dim3(b1), dim3(128), lds, p, actions, fill);
        else
          TE_LAUNCH((substeps_kernel<F, kNoise, decltype(fill_c)::value, false, kNoise>), dim3(b1), dim3(128), lds, p, actions, fill);
      } else if (p.cfg.control_every_substep)
        TE_LAUNCH((substeps_kernel<F, kNoise, decltype(fill_c)::value, true>), dim3(b1), dim3(TE_K1_BLOCK), 0, p, actions, fill);
      else
        TE_LAUNCH((substeps_kernel<F, kNoise, decltype(fill_c)::value, false>), dim3(b1), dim3(TE_K1_BLOCK), 0, p, actions, fill);
    };
    if (noise) { if (fill.lidar) go(std::true_type{}, std::true_type{}); else go(std::true_type{}, std::false_type{}); }
    else { if (fill.lidar) go(std::false_type{}, std::true_type{}); else go(std::false_type{}, std::false_type{}); }
  });
  if (prof && !prof_ext) { TE_HIP(hipEventRecord(pev[1], st)); TE_HIP(hipEventRecord(pev[2], st)); }
  if (prof_ext) { ev_a = pev[2]; ev_b = stack ? nullptr : pev[3]; }
  const int b2 = (p.N + kEPB - 1) / kEPB;
  // the engage/observe kernel patches the agent's own sphere only in the classic layout; in stacked mode all LIDAR
  // output comes from stacked_kernel
  StepOut o{reward, done, info, ObsOut{stack ? nullptr : obs_lidar, obs_inertial, obs_last_action},
            ObsOut{stack ? nullptr : terminal_lidar, terminal_inertial, terminal_last_action}, e->prev_cells, stack ? 0 : persist};
  const bool contact = p.cfg.drone_contact != 0;   // its own instantiations: the contact pass would cost every launch ~150 VGPRs
  if (e->engage_slots == 2 && p.D > 32) TE_LAUNCH((engage_slots_multi_kernel<3, false, true>), dim3(b2), dim3(64 * e->slot_waves), e->slot_lds, p, actions, o);
  else if (e->engage_slots == 2 && !stack) TE_LAUNCH((engage_slots_multi_kernel<2, true>), dim3(b2), dim3(64 * e->slot_waves), e->slot_lds, p, actions, o);
  else if (e->engage_slots == 2 && e->slot_spw == 1) TE_LAUNCH((engage_slots_multi_kernel<1, false>), dim3(b2), dim3(64 * e->slot_waves), e->slot_lds, p, actions, o);
  else if (e->engage_slots == 2 && e->slot_spw == 3) TE_LAUNCH((engage_slots_multi_kernel<3, false>), dim3(b2), dim3(64 * e->slot_waves), e->slot_lds, p, actions, o);
  else if (e->engage_slots == 2) TE_LAUNCH((engage_slots_multi_kernel<2, false>), dim3(b2), dim3(64 * e->slot_waves), e->slot_lds, p, actions, o);
  else if (e->engage_slots && e->family == FAM_STAGE02) TE_LAUNCH((engage_slots_stage02_kernel<kSlotWaves>), dim3(b2), dim3(64 * p.D), (size_t)slot_lds_rows(p.D, p.cfg.n_pursuers) * 256, p, actions, o);
  else if (e->engage_slots && !contact && e->slot_wpe8) TE_LAUNCH((engage_slots_kernel<kSlotWaves, 8>), dim3(b2), dim3(64 * p.D), (size_t)slot_lds_rows(p.D, p.cfg.n_pursuers) * 256, p, actions, o);
  else if (e->engage_slots && !contact) TE_LAUNCH((engage_slots_kernel<kSlotWaves>), dim3(b2), dim3(64 * p.D), (size_t)slot_lds_rows(p.D, p.cfg.n_pursuers) * 256, p, actions, o);
  else if (e->engage_regs == 1 && !contact) TE_LAUNCH((engage_kernel<2, 9>), dim3(b2), dim3(64), 0, p, actions, o);
  else if (e->engage_regs == 2 && !contact) TE_LAUNCH((engage_kernel<6, 12>), dim3(b2), dim3(64), 0, p, actions, o);
#ifndef TE_DEBUG_STAMPS  // the stamp build leaves the contact variants out (the compiler rejects them next to the stamp stores)
  else if (e->engage_regs == 1) TE_LAUNCH((engage_kernel<2, 9, true>), dim3(b2), dim3(64), 0, p, actions, o);
  else if (e->engage_regs == 2) TE_LAUNCH((engage_kernel<6, 12, true>), dim3(b2), dim3(64), 0, p, actions, o);
#endif
  else if (e->engage_regs == 5) TE_LAUNCH((engage_kernel<7, 30>), dim3(b2), dim3(64), 0, p, actions, o);
  else if (e->engage_regs == 3) TE_LAUNCH((engage_stage02_kernel<2, 8>), dim3(b2), dim3(64), 0, p, actions, o);
  else if (e->engage_regs == 4) TE_LAUNCH(engage_stage01_kernel, dim3(b2), dim3(64), 0, p, actions, o);
  else launch_by_family(e->family, [&](auto fam) {
    if (e->k2_threads == 512) TE_LAUNCH((engage_observe_kernel<FAM_LEVEL4, 512>), dim3(b2), dim3(512), e->lds_bytes, p, actions, o);
    else TE_LAUNCH((engage_observe_kernel<decltype(fam)::value>), dim3(b2), dim3(256), e->lds_bytes, p, actions, o);
  });
  if (stack) {
    ev_a = nullptr; ev_b = nullptr;
    // the first launch pushes this step's ring entries (all wingmen) and serves observer 0; te_step_students adds one launch per further wingman
    if (e->stack_regs) {  // one wave per (chunk, wingman) pushes this step's ring entries, then one 5-wave workgroup per chunk and observer
      StackParams sp{p.cfg, p.snap, p.ring, p.N, p.Npad, p.D, p.entry_words, 1, 0, n_obs, persist, e->prev_cells};
      const unsigned push_waves = (unsigned)b2 * (unsigned)p.cfg.n_pursuers;
      if (e->stack_regs == 18 && e->push_split) TE_LAUNCH((ring_push_kernel<18, kPushSplit>), dim3(push_waves), dim3(64 * kPushSplit), (size_t)4 * 18 * 256, sp);
      else if (e->stack_regs == 18) TE_LAUNCH((ring_push_kernel<18, 1>), dim3(push_waves), dim3(64), 0, sp);
      else TE_LAUNCH((ring_push_kernel<37, 1>), dim3(push_waves), dim3(64), 0, sp);
      for (int ob = 0; ob < n_obs; ++ob) {
        sp.push = ob == 0 ? 1 : 0; sp.observer = ob;   // the first view clears the ring of auto-reset envs when it is through
        if (prof_ext && ob == n_obs - 1) ev_b = pev[3];
        if (e->stack_regs == 18) TE_LAUNCH((stack_view_kernel<18>), dim3(b2), dim3(kViewThreads), e->view_lds_bytes, sp, *stack);
        else TE_LAUNCH((stack_view_kernel<37>), dim3(b2), dim3(kViewThreads), e->view_lds_bytes, sp, *stack);
      }
    } else
    for (int ob = 0; ob < n_obs; ++ob) {
      StackParams sp{p.cfg, p.snap, p.ring, p.N, p.Npad, p.D, p.entry_words, ob == 0 ? 1 : 0, ob, n_obs, 0, nullptr};
      if (prof_ext && ob == n_obs - 1) ev_b = pev[3];
      TE_LAUNCH(stacked_kernel, dim3(b2), dim3(kStackThreads), e->stack_lds_bytes, sp, *stack);
    }
  }
#undef TE_LAUNCH
  e->last_stacked = persist ? obs_lidar : nullptr; e->last_n_obs = n_obs;
  if (prof) { if (!prof_ext) TE_HIP(hipEventRecord(pev[3], st)); e->prof_used += 4; }
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_step(te_env* e, const float* actions, float* obs_lidar, float* obs_inertial,
                                                   float* obs_last_action, float* reward, uint8_t* done, int32_t* info,
                                                   float* terminal_lidar, float* terminal_inertial, float* terminal_last_action,
                                                   void* stream) {
  if (host_io(e)) {
    if (!actions || !reward || !done || !info) return fail("te_step: actions, reward, done and info are required");
    DeviceGuard guard(e->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)e->p.N, lw = (size_t)lidar_words(e->p.cfg) * 4;
    const te_env::HostStage& h = e->hs;
    TE_H2D(h.actions, actions, N * 16);
    if (step_impl(e, h.actions, obs_lidar ? h.lidar : nullptr, lw / 4, obs_inertial ? h.inertial : nullptr, obs_last_action ? h.last_action : nullptr,
                  h.reward, h.done, h.info, terminal_lidar ? h.t_lidar : nullptr, terminal_inertial ? h.t_inertial : nullptr,
                  terminal_last_action ? h.t_last_action : nullptr, nullptr, stream)) return 1;
    TE_D2H(obs_lidar, h.lidar, N * lw); TE_D2H(obs_inertial, h.inertial, N * 60); TE_D2H(obs_last_action, h.last_action, N * 16);
    TE_D2H(reward, h.reward, N * 4); TE_D2H(done, h.done, N); TE_D2H(info, h.info, N * 16);
    TE_HIP(hipStreamSynchronize(st));
    // terminal rows are defined only for envs that are done (include/threatengage.h): ~1 % of the envs per step.  Copying the three
    // buffers whole doubled the PCIe bytes of a step (6.5 instead of 12.5 M env-steps/s at 65 536 envs, tools/host_io_bench.py): the done
    // rows are gathered on the device (into the observation staging, whose copy has landed), cross in one piece each and are
    // scattered into the caller's arrays here.
    if (terminal_lidar || terminal_inertial || terminal_last_action) {
      std::vector<int32_t>& idx = e->done_idx;
      idx.clear();
      for (size_t i = 0; i < N; ++i) if (done[i]) idx.push_back((int32_t)i);
      const size_t n = idx.size();
      if (n) {
        const size_t row[3] = {terminal_lidar ? lw : 0, terminal_inertial ? (size_t)60 : 0, terminal_last_action ? (size_t)16 : 0};
        const float* src[3] = {h.t_lidar, h.t_inertial, h.t_last_action};
        // (with the persistent observation the LIDAR staging IS the observation the next step patches in place: compact elsewhere)
        float* tmp[3] = {e->persist_on && h.t_gather ? h.t_gather : h.lidar, h.inertial, h.last_action};
        if (row[0] && tmp[0] == h.lidar) e->last_stacked = nullptr;   // the staging no longer holds the observation: next step is dense
        float* dst[3] = {terminal_lidar, terminal_inertial, terminal_last_action};
        e->done_rows.resize(n * (row[0] + row[1] + row[2]));
        TE_HIP(hipMemcpyAsync(h.info, idx.data(), n * 4, hipMemcpyHostToDevice, st));   // the info staging is free again too
        char* land = e->done_rows.data();
        for (int k = 0; k < 3; ++k) {
          if (!row[k]) continue;
          hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(64), 0, st, src[k], tmp[k], h.info, (int)(row[k] / 4));
          TE_HIP(hipMemcpyAsync(land, tmp[k], n * row[k], hipMemcpyDeviceToHost, st));
          land += n * row[k];
        }
        TE_HIP(hipStreamSynchronize(st));
        land = e->done_rows.data();
        for (int k = 0; k < 3; ++k) {
          if (!row[k]) continue;
          for (size_t j = 0; j < n; ++j) memcpy((char*)dst[k] + (size_t)idx[j] * row[k], land + j * row[k], row[k]);
          land += n * row[k];
        }
      }
    }
    return 0;
  }
  return step_impl(e, actions, obs_lidar, e ? (size_t)lidar_words(e->p.cfg) : 0, obs_inertial, obs_last_action, reward, done, info, terminal_lidar,
                   terminal_inertial, terminal_last_action, nullptr, stream);
}

__attribute__((visibility("default"))) int te_step_stacked(te_env* e, const float* actions, float* obs_stacked, uint8_t* obs_mask,
                                                           float* obs_inertial, float* obs_last_action, float* reward, uint8_t* done,
                                                           int32_t* info, float* terminal_stacked, uint8_t* terminal_mask,
                                                           float* terminal_inertial, float* terminal_last_action, void* stream) {
  if (!obs_stacked || !obs_mask) return fail("te_step_stacked: obs_stacked and obs_mask are required");
  if (((uintptr_t)obs_stacked & 15) || (terminal_stacked && ((uintptr_t)terminal_stacked & 15)))
    return fail("te_step_stacked: the stacked buffers must be 16-byte aligned");
  if ((terminal_stacked == nullptr) != (terminal_mask == nullptr)) return fail("te_step_stacked: terminal_stacked and terminal_mask go together");
  const StackOut so{obs_stacked, obs_mask, terminal_stacked, terminal_mask};
  return step_impl(e, actions, obs_stacked, TE_OBS_STACKED_WORDS, obs_inertial, obs_last_action, reward, done, info, nullptr,
                   terminal_inertial, terminal_last_action, &so, stream);
}

// Persistent observation (opt-in).  The caller promises that nobody but this library writes the observation buffer it passes (obs_lidar
// of te_step; obs_stacked of te_step_stacked / te_step_students / te_observe_stacked).  While it keeps passing the SAME buffer, a step
// rewrites only the cells that change: the cells the previous observation patched (<= D - 1 per sphere, recorded by the kernel that patched
// them) go back to one — by erase waves of the sub-step launch, next to its flights — and the new ones are patched, instead of streaming
// the whole background (4 KB per env for the own sphere, 24 KB for the stacked one) and patching it.  The buffer's content is bit for bit
// what the dense path writes (tests/test_gpu_level5.py, tests/test_gpu_properties.py).  A different pointer (a rollout buffer that
// advances every step) falls back to the dense path for that call.  Terminal buffers are always written densely (done envs only).
// Served by the register kernels (engage_kernel / engage_stage0x_kernel, stack_view_kernel); TE_ENGAGE=lds / TE_STACKED=lds ignore it.
__attribute__((visibility("default"))) int te_set_persistent_obs(te_env* e, int32_t on) {
  if (!e) return fail("te_set_persistent_obs: null env");
  DeviceGuard guard(e->device);
  e->last_stacked = nullptr;
  e->persist_on = on != 0 && (e->p.ring ? e->stack_regs != 0 : e->engage_regs != 0);   // the register kernels record what they patch; the LDS fallbacks stay dense
  if (e->persist_on && !e->prev_cells) {
    const int observers = !e->p.ring ? 1 : all_scripted(e->p.cfg) ? e->p.cfg.n_pursuers : 1;   // te_step_students serves every wingman
    const size_t bytes = (size_t)observers * (e->p.ring ? TE_STACK_SPHERES : 1) * (size_t)e->p.D * (size_t)e->p.Npad * sizeof(uint16_t);
    if (hipMalloc(&e->prev_cells, bytes) != hipSuccess) { e->prev_cells = nullptr; e->persist_on = false; return fail("te_set_persistent_obs: out of device memory"); }
    e->prev_observers = observers;   // (no clear: the first call through a buffer always records before anything erases)
  }
  if (e->persist_on && host_io(e) && !e->hs.t_gather) {
    if (hipMalloc(&e->hs.t_gather, (size_t)e->p.N * lidar_words(e->p.cfg) * 4) != hipSuccess) {
      e->hs.t_gather = nullptr; e->persist_on = false; return fail("te_set_persistent_obs: out of device memory (terminal-row staging)");
    }
  }
  return 0;
}

// Level5DumbMultiObs.compute_info rows of every pursuer after the step (level5_dumb_multiobs.py:116-150): normalised IMU + gun state,
// the behaviour tree's command of this step as an action (unit direction, speed: loyalwingman_navigator.py:301,325,350), armed flag
__global__ __launch_bounds__(256) void students_rows_kernel(Params p, const uint8_t* __restrict__ done, float* __restrict__ inertial,
                                                            float* __restrict__ last_action, uint8_t* __restrict__ active) {
  const int i = blockIdx.x * 256 + threadIdx.x, P = p.cfg.n_pursuers;
  if (i >= p.N * P) return;
  const int s = i / p.N, env = i - s * p.N;   // slot-major: consecutive threads = consecutive envs of one plane
  const GView v{p.dstate, p.estate, p.D, p.Npad, env, P};
  const size_t row = (size_t)env * P + s;
  float in[TE_OBS_INERTIAL_WORDS];
  inertial_obs(p.cfg, v, v.egi(TE_E_STEP), in, s);
#pragma unroll
  for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) inertial[row * TE_OBS_INERTIAL_WORDS + k] = in[k];
  const bool armed = v.gi(TE_D_ARMED, s) != 0;
  const bool fresh = done[env] != 0 && p.cfg.auto_reset;   // the reset observation: nobody has been commanded yet
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (armed && !fresh) {
    const float sp = p.cfg.ally_speed;
    a = make_float4(v.gf(TE_D_SETPOINT, s) / sp, v.gf(TE_D_SETPOINT + 1, s) / sp, v.gf(TE_D_SETPOINT + 3, s) / sp, sp);
  }
  reinterpret_cast<float4*>(last_action)[row] = a;
  active[row] = armed ? 1 : 0;
}

__attribute__((visibility("default"))) int te_step_students(te_env* e, float* stacked, uint8_t* mask, float* inertial, float* last_action,
                                                            uint8_t* active, float* reward, uint8_t* done, int32_t* info, void* stream) {
  if (!e) return fail("te_step_students: null env");
  if (!e->p.ring || !all_scripted(e->p.cfg)) return fail("te_step_students: needs cfg.stacked_obs and an all-scripted task (cfg.agent_scripted or cfg.evaluation)");
  if (!stacked || !mask || !inertial || !last_action || !active || !reward || !done || !info) return fail("te_step_students: every output buffer is required");
  if (((uintptr_t)stacked & 15) || ((uintptr_t)last_action & 15) || ((uintptr_t)info & 15)) return fail("te_step_students: stacked, last_action and info must be 16-byte aligned");
  if (!e->zero_actions) return fail("te_step_students: internal error: no action buffer");
  const int P = e->p.cfg.n_pursuers;
  const StackOut so{stacked, mask, nullptr, nullptr};
  if (step_impl(e, e->zero_actions, stacked, (size_t)P * TE_OBS_STACKED_WORDS, nullptr, nullptr, reward, done, info, nullptr, nullptr, nullptr, &so, stream, P)) return 1;
  DeviceGuard guard(e->device);
  const int n = e->p.N * P;
  hipLaunchKernelGGL(students_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->p, done, inertial, last_action, active);
  TE_HIP(hipGetLastError());
  return 0;
}

// te_observe_stacked beyond 32 drones per env: the agent's inertial row (normalize_inertial_data + gun state) and last action, one lane per env
__global__ __launch_bounds__(256) void agent_rows_kernel(Params p, float* __restrict__ inertial, float* __restrict__ last_action) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= p.N) return;
  const GView v{p.dstate, p.estate, p.D, p.Npad, env, p.cfg.n_pursuers};
  if (inertial) {
    float in[TE_OBS_INERTIAL_WORDS];
    inertial_obs(p.cfg, v, v.egi(TE_E_STEP), in, 0);
#pragma unroll
    for (int k = 0; k < TE_OBS_INERTIAL_WORDS; ++k) inertial[(size_t)env * TE_OBS_INERTIAL_WORDS + k] = in[k];
  }
  if (last_action)
    reinterpret_cast<float4*>(last_action)[env] = make_float4(v.egf(TE_E_LAST_ACTION), v.egf(TE_E_LAST_ACTION + 1), v.egf(TE_E_LAST_ACTION + 2), v.egf(TE_E_LAST_ACTION + 3));
}

__attribute__((visibility("default"))) int te_observe_stacked(te_env* e, float* obs_stacked, uint8_t* obs_mask, float* obs_inertial,
                                                              float* obs_last_action, void* stream) {
  if (!e) return fail("te_observe_stacked: null env");
  if (!e->p.ring) return fail("te_observe_stacked: this te_env was created without cfg.stacked_obs");
  if (e->p.D > kMaxD && all_scripted(e->p.cfg)) return fail("te_observe_stacked: an all-scripted task with more than 32 drones per env: the observation comes out of te_step_students");
  if (!obs_stacked || !obs_mask || ((uintptr_t)obs_stacked & 15)) return fail("te_observe_stacked: obs_stacked (16-byte aligned) and obs_mask are required");
  DeviceGuard guard(e->device);
  hipStream_t st = (hipStream_t)stream;
  const Params& p = e->p;
  const int blocks = (p.N + kEPB - 1) / kEPB;
  hipLaunchKernelGGL(fill_ones_kernel, dim3(2048), dim3(256), 0, st, obs_stacked, (size_t)p.N * TE_OBS_STACKED_WORDS);
  hipLaunchKernelGGL(snapshot_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, p);
  if (p.D > kMaxD) {  // Level5FusionTask (36 drones): observe_kernel stages at most 32 slots; the two rows need none of them
    if (obs_inertial || obs_last_action) hipLaunchKernelGGL(agent_rows_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, p, obs_inertial, obs_last_action);
  } else {
    hipLaunchKernelGGL(observe_kernel, dim3(blocks), dim3(256), e->lds_bytes, st, p, ObsOut{nullptr, obs_inertial, obs_last_action});
  }
  const int persist = (e->persist_on && e->stack_regs && e->prev_observers >= 1) ? 2 : 0;   // dense fill above; the cells are recorded for the next step
  StackParams sp{p.cfg, p.snap, p.ring, p.N, p.Npad, p.D, p.entry_words, 0, 0, 1, persist, e->prev_cells};
  e->last_stacked = persist ? obs_stacked : nullptr; e->last_n_obs = 1;
  if (e->stack_regs == 18) hipLaunchKernelGGL((stack_view_kernel<18>), dim3(blocks), dim3(kViewThreads), e->view_lds_bytes, st, sp, StackOut{obs_stacked, obs_mask, nullptr, nullptr});
  else if (e->stack_regs == 37) hipLaunchKernelGGL((stack_view_kernel<37>), dim3(blocks), dim3(kViewThreads), e->view_lds_bytes, st, sp, StackOut{obs_stacked, obs_mask, nullptr, nullptr});
  else hipLaunchKernelGGL(stacked_kernel, dim3(blocks), dim3(kStackThreads), e->stack_lds_bytes, st, sp, StackOut{obs_stacked, obs_mask, nullptr, nullptr});
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_random_actions(te_env* e, float* actions, uint64_t seed, uint64_t step_index, void* stream) {
  if (!e || !actions) return fail("te_random_actions: null argument");
  DeviceGuard guard(e->device);
  hipStream_t st = (hipStream_t)stream;
  float* dst = host_io(e) ? e->hs.actions : actions;
  if ((uintptr_t)dst & 15) return fail("te_random_actions: actions must be 16-byte aligned");
  hipLaunchKernelGGL(random_actions_kernel, dim3((e->p.N + 255) / 256), dim3(256), 0, st, e->p, dst, seed, step_index);
  TE_HIP(hipGetLastError());
  if (host_io(e)) { TE_D2H(actions, dst, (size_t)e->p.N * 16); TE_HIP(hipStreamSynchronize(st)); }
  return 0;
}

static size_t ring_words(const te_env* e) {  // level5: the snapshot ring rides at the end of the state blob
  return e->p.ring ? (size_t)e->p.N * e->p.cfg.n_pursuers * TE_RING_DEPTH * (size_t)e->p.entry_words : 0;
}
__attribute__((visibility("default"))) int te_state_words(const te_env* e, size_t* out_words) {
  if (!e || !out_words) return fail("te_state_words: null argument");
  *out_words = (size_t)e->p.N * ((size_t)e->p.D * TE_DRONE_WORDS + TE_ENV_WORDS) + ring_words(e);
  return 0;
}

__attribute__((visibility("default"))) int te_get_state(te_env* e, void* dst_device, size_t words, void* stream) {
  size_t need = 0;
  if (te_state_words(e, &need)) return 1;
  if (!dst_device || words != need) return fail("te_get_state: buffer must hold exactly te_state_words() words");
  DeviceGuard guard(e->device);
  if (host_io(e)) {   // dst is a host buffer
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(planes_to_blob, dim3(1024), dim3(256), 0, st, e->p, e->hs.blob);
    TE_HIP(hipGetLastError());
    TE_D2H(dst_device, e->hs.blob, need * 4);
    TE_HIP(hipStreamSynchronize(st));
    return 0;
  }
  hipLaunchKernelGGL(planes_to_blob, dim3(1024), dim3(256), 0, (hipStream_t)stream, e->p, (uint32_t*)dst_device);
  if (e->p.ring)
    TE_HIP(hipMemcpyAsync((uint32_t*)dst_device + (need - ring_words(e)), e->p.ring, ring_words(e) * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  TE_HIP(hipGetLastError());
  return 0;
}

__attribute__((visibility("default"))) int te_set_state(te_env* e, const void* src_device, size_t words, void* stream) {
  size_t need = 0;
  if (te_state_words(e, &need)) return 1;
  if (!src_device || words != need) return fail("te_set_state: buffer must hold exactly te_state_words() words");
  DeviceGuard guard(e->device);
  if (host_io(e)) {   // src is a host buffer
    hipStream_t st = (hipStream_t)stream;
    TE_H2D(e->hs.blob, src_device, need * 4);
    TE_HIP(hipStreamSynchronize(st));   // the caller may reuse or free its (possibly pinned) blob as soon as the call returns
    src_device = e->hs.blob;
  }
  hipLaunchKernelGGL(blob_to_planes, dim3(1024), dim3(256), 0, (hipStream_t)stream, e->p, (const uint32_t*)src_device);
  if (e->p.ring)
    TE_HIP(hipMemcpyAsync(e->p.ring, (const uint32_t*)src_device + (need - ring_words(e)), ring_words(e) * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (e->family == FAM_LEVEL4)
    hipLaunchKernelGGL(prepare_commands_kernel, dim3((e->p.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->p);
  launch_census(e, (hipStream_t)stream);
  TE_HIP(hipGetLastError());
  return 0;
}

// Kernel timing with HIP events on the stream te_step launches on.
__attribute__((visibility("default"))) int te_profile_begin(te_env* e, int32_t max_steps) {
  if (!e || max_steps < 1) return fail("te_profile_begin: bad argument");
  DeviceGuard guard(e->device);
  while ((int)e->events.size() < 4 * max_steps) {
    hipEvent_t ev;
    TE_HIP(hipEventCreate(&ev));
    e->events.push_back(ev);
  }
  const char* mode = getenv("TE_PROF");
  e->prof_markers = mode && !strcmp(mode, "markers");
  e->prof_cap = 4 * max_steps;
  e->prof_used = 0;
  return 0;
}
// Averages over the steps recorded since te_profile_begin; blocks until they have finished.
__attribute__((visibility("default"))) int te_profile_end(te_env* e, float* substeps_ms, float* engage_observe_ms, int32_t* n_steps) {
  if (!e) return fail("te_profile_end: null env");
  DeviceGuard guard(e->device);
  const int n = e->prof_used / 4;
  double a = 0, b = 0;
  if (n > 0) TE_HIP(hipEventSynchronize(e->events[e->prof_used - 1]));
  for (int i = 0; i < n; ++i) {
    float m1 = 0, m2 = 0;
    TE_HIP(hipEventElapsedTime(&m1, e->events[4 * i], e->events[4 * i + 1]));
    TE_HIP(hipEventElapsedTime(&m2, e->events[4 * i + 2], e->events[4 * i + 3]));
    a += m1; b += m2;
  }
  if (substeps_ms) *substeps_ms = n ? (float)(a / n) : 0.0f;
  if (engage_observe_ms) *engage_observe_ms = n ? (float)(b / n) : 0.0f;
  if (n_steps) *n_steps = n;
  e->prof_cap = 0; e->prof_used = 0;
  return 0;
}

// Diagnostic builds (-DTE_DEBUG_STAMPS): s_memrealtime (100 MHz) stamps one workgroup wrote at its phase
// boundaries during the last launch; all zeros in a normal build.
__attribute__((visibility("default"))) int te_debug_stamps(te_env* e, uint64_t* out_host, int32_t n) {
  if (!e || !out_host || n < 1 || (size_t)n > (e->dbg_words ? e->dbg_words : 64 + 16 * (size_t)(e->p.Npad / kEPB + 1))) return fail("te_debug_stamps: bad argument");
  memset(out_host, 0, (size_t)n * sizeof(uint64_t));
  if (!e->p.dbg) return 0;
  DeviceGuard guard(e->device);
  TE_HIP(hipDeviceSynchronize());
  TE_HIP(hipMemcpy(out_host, e->p.dbg, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
