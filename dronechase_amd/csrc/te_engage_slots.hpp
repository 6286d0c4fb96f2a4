// te_engage_slots.hpp — the engage/observe step with ONE WAVE PER (chunk of 64 envs, drone slot): the small-shard form.
//
// engage_kernel (te_engage.hpp) gives every env one lane and every chunk one wave: ~2 000 straight-line VALU instructions per wave,
// 11-17 us of one dependent chain, which is the whole kernel when a shard has fewer chunks than the chip has SIMDs (8 192 envs per GPU =
// 128 waves for 1 024 SIMDs: the metric's 8-GPU shard).  Here a chunk is a workgroup of D waves; lane = env as before, so every plane
// access stays one coalesced 256-byte row and "the other drones of my env" are the same lane of another wave's LDS row:
//
//   P1  wave s: its own slot's IMU position + armed flag, the agent's pose, the env counters (one round of independent loads; a pursuer
//       wave also requests every live invader's row: its targeting needs no other wave) -> |p|, dome / origin flags, the LIDAR cell and
//       range of its drone in the agent's frame; a pursuer wave: closest invader, shot (Philox only on the lanes that fire), explosion
//       -> one LDS row each (flags, position, cell, range, pursuer record)                                              -> barrier 1
//   P3  every wave, redundantly (a few dozen LDS reads): masks, kills, counters, termination, the next round / auto-reset decision;
//       wave s alone: Quadcopter.disarm of its drone, closer-wins ownership of its LIDAR cell (the minimum of (range, slot) over the
//       armed drones in the cell — the fixed point of the slot-order loop of engage_kernel), its three patches, its slot's respawn
//       (every lane that needs one draws its own Philox words: no env-by-env loop); wave 0 alone: reward, env record, outputs
//                                                                                                                      -> barrier 2
//   P4  wave 0: the [64, 15] rows; pursuer waves: TE_X_REF and the behaviour tree's command on the post-spawn positions; every wave:
//       its slot's share of the next flight plan, the persistent-observation record, the patches of auto-reset envs' terminal tiles.
//
// Bit for bit what engage_kernel<2, 9> writes (tests/test_gpu_engage_slots.py): the same device functions on the same operands in the
// same order per value.  Serves the level4 family without a snapshot ring (exp02/03/04/05, evaluation) for D <= 16, drone_contact off;
// selected per te_env by shard size (te_create) or TE_ENGAGE=slots / regs.
#pragma once
#include "te_engage.hpp"

namespace te {

constexpr int kSlotWaves = 16;     // waves of a chunk's workgroup = drone slots served (1 024 threads)
constexpr int kSlotPursuers = 4;   // pursuers served (the exp tasks have 2)
// te_create's default: every shard size takes the slot waves since dead-slot waves retire early (round 4: stage03 x 65 536, driver's window
// 18.4 vs 21.6 us, steady state and all-armed a tie; x 32 768: 14.2 vs 20.1; x 8 192: 10.4 vs 16.9 us; profiles/r04_j_ab_engage_slots_v4.txt);
// TE_ENGAGE=regs brings engage_kernel back, and a te_env above this many (env, slot) pairs keeps it
constexpr long long kSlotsMaxPairs = 1ll << 40;

struct SlotRows {  // LDS rows of 64 words
  int D, P;
  // accumulator rows, zeroed before barrier 0 and OR-ed into by every wave (bit s = slot s): no D-long gather loops afterwards
  TE_DEV int accS() const { return 0; }      // armed & valid
  TE_DEV int accZone() const { return 1; }   // armed and outside the dome
  TE_DEV int accOrg() const { return 2; }    // armed and inside the origin range
  TE_DEV int accOwn() const { return 3; }    // owns its LIDAR cell (terminal envs alike)
  TE_DEV int pos(int k, int s) const { return 4 + k * D + s; }        // IMU position as loaded
  TE_DEV int cellr(int s) const { return 4 + 3 * D + 2 * s; }         // two rows per slot: lane's {cell, range} as one 8-byte word
  TE_DEV int npos(int k, int s) const { return 4 + (5 + k) * D + s; } // position after the spawn
  TE_DEV int prec(int q) const { return 4 + 8 * D + q; }              // pursuer record: target + 1 | HIT | EXPLODE | SUICIDE
  TE_DEV int prec2(int q) const { return 4 + 8 * D + 2 * q; }         // stage02: two rows per pursuer: lane's {record, distance to the target}
};
__host__ __device__ inline int slot_lds_rows(int D, int P) { return 4 + 8 * D + 2 * P; }
enum : uint32_t { SLOT_HIT = 1u << 8, SLOT_EXPLODE = 1u << 9, SLOT_SUICIDE = 1u << 10 };

#define TE_SLOT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#if defined(TE_DEBUG_STAMPS)
// phase stamps (diagnostic builds only; tools/slots_stamps.py): waves 0 and 1 of every workgroup, idx 0..7 each
#define TE_WSTAMP(idx, wait)                                                                                              \
  do {                                                                                                                    \
    if (wait) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
    if (p.dbg && s < 2 && lane == 0) p.dbg[64 + blockIdx.x * 16 + s * 8 + (idx)] = __builtin_amdgcn_s_memrealtime();     \
  } while (0)
#else
#define TE_WSTAMP(idx, wait) do {} while (0)
#endif

// WPE: waves per SIMD the register budget is held to.  6 (75 VGPRs) by default; 8 (64 VGPRs, 24 B of scratch; TE_SLOT_WPE8=1) keeps one more
// workgroup per CU in the heavy regimes of a large shard (steady-state engage 29.5 -> 26.2 us at 65 536 envs) and costs the light ones 1 us.
template <int DM, int WPE = 6>
__global__ __launch_bounds__(DM * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE > 6 ? WPE : 10))) void engage_slots_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  TE_EXACT
  extern __shared__ uint32_t sm[];
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers;
  const SlotRows R{D, P};
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // this wave's drone slot
  const int env = blockIdx.x * 64 + lane;
  const bool valid = env < p.N;
  const GView g{p.dstate, p.estate, D, p.Npad, env, P};
  const EnvIO io(p, env);
  const uint32_t pur_bits = (1u << P) - 1u, all_bits = (1u << D) - 1u, inv_bits = all_bits & ~pur_bits;
  const bool scripted = all_scripted(c);
  const bool is_p = s < P;
  auto L = [&](int row) -> uint32_t& { return sm[row * 64 + lane]; };
  auto Lf = [&](int row) { return __uint_as_float(sm[row * 64 + lane]); };
  auto Lor = [&](int row, uint32_t v) { __hip_atomic_fetch_or(&sm[row * 64 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  TE_WSTAMP(0, 0);

  // ---- P0: one round of independent loads, all requested before anything is looked at ----------------------------------------------------
  const uint32_t* __restrict__ lm32 = reinterpret_cast<const uint32_t*>(p.live_mask);
  const uint32_t live = (uint32_t)__builtin_amdgcn_readfirstlane(lm32[2 * blockIdx.x]) & all_bits;   // D <= 16
  const bool mine_live = is_p || ((live >> s) & 1u);
  // A slot nobody of the chunk has armed AND that no env of the chunk can arm in this step (an env arms the invaders of its NEXT round, or
  // of round 1 after a reset) has nothing to do: its wave retires on one plane row, before the first barrier (s_barrier does not wait for
  // ended waves), and leaves its SIMD slot to the next workgroup — in a rollout 5-8 of stage03's 11 slots are live.  The one word such a
  // slot would get, NAV_STATE = Wait in the envs that start a round (spawn_slot_at), is written by wave 0.
  auto next_round_of = [&](int r) { return r + (r < c.n_rounds ? 1 : c.n_rounds); };   // advance_round (exp03_vFinal_task.py:155-175)
  auto may_be_spawned = [&](int slot, int r) { const int i = slot - P; return i < invaders_in_round(c, next_round_of(r)) || i < invaders_in_round(c, 1); };
  if (!mine_live) {
    const int r = (int)io.le(TE_E_ROUND);
    if (__ballot(valid && may_be_spawned(s, r)) == 0ull) return;
  }
  // (a slot nobody of the chunk has armed is requested all the same: a conditional request makes the compiler wait for it inside the branch,
  // in front of every later request; its lanes are masked below)
  float mx = io.ldf(TE_D_OBS_POS, s), my = io.ldf(TE_D_OBS_POS + 1, s), mz = io.ldf(TE_D_OBS_POS + 2, s);
  const uint32_t marmed_w = io.ld(TE_D_ARMED, s);
  const V3 apos{io.ldf(TE_D_OBS_POS, 0), io.ldf(TE_D_OBS_POS + 1, 0), io.ldf(TE_D_OBS_POS + 2, 0)};
  float ag[9];  // OBS_EULER, OBS_VEL, OBS_RATE of the agent (wave 0 needs all nine, the others the attitude)
#pragma unroll
  for (int k = 0; k < 3; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
#pragma unroll
  for (int k = 3; k < 9; ++k) ag[k] = 0.0f;
  const uint32_t w_step = io.le(TE_E_STEP), w_max_step = io.le(TE_E_MAX_STEP), w_round = io.le(TE_E_ROUND), w_episode = io.le(TE_E_EPISODE);
  // wave 0: the rest of the env record and the action
  uint32_t w_last_dist = 0u, w_ak = 0u, w_lk = 0u, w_dd = 0u;
  float4 act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (s == 0) {
#pragma unroll
    for (int k = 3; k < 9; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
    w_last_dist = io.le(TE_E_LAST_DIST); w_ak = io.le(TE_E_AGENT_KILLS); w_lk = io.le(TE_E_ALLIES_KILLS); w_dd = io.le(TE_E_DEADS);
    if (valid) act = reinterpret_cast<const float4*>(actions)[env];
  }
  // pursuer waves: gun and formation point (their targeting reads the invaders' rows from LDS behind barrier 1: no per-wave register arrays,
  // 64 VGPRs = eight waves per SIMD, two workgroups of eleven waves per CU)
  uint32_t w_mun = 0u, w_lf = 0u; float fx = 0.0f, fy = 0.0f, fz = 0.0f;
  if (is_p) {
    w_mun = io.ld(TE_D_MUNITION, s); w_lf = io.ld(TE_D_LAST_FIRED, s);
    fx = io.ldf(TE_D_FORMATION, s); fy = io.ldf(TE_D_FORMATION + 1, s); fz = io.ldf(TE_D_FORMATION + 2, s);
  }
  // an opaque zero, defined HERE: a comparison against it cannot be scheduled in front of this line.  (Comparing a requested word with the
  // literal 0 lets the compiler sink the test, and the wait for the word, into the branch that requested it: every live invader's row then
  // cost its own memory round trip, 1.9 us of targeting per pursuer wave.)
  uint32_t zero = 0u;
  asm volatile("" : "+s"(zero));
  if (s == 0) { L(R.accS()) = 0u; L(R.accZone()) = 0u; L(R.accOrg()) = 0u; L(R.accOwn()) = 0u; }
  TE_SLOT_BARRIER();   // barrier 0: the accumulator rows are zero
  int step = (int)w_step + 1;  // AGENT_STEP_BROADCAST (exp03_vFinal_environment.py:177-182)
  int max_step = (int)w_max_step, round = (int)w_round;
  uint32_t episode = w_episode;
  int mun = (int)w_mun, lf = (int)w_lf;
  const float last_dist = __uint_as_float(w_last_dist);
  int agent_kills = (int)w_ak, allies_kills = (int)w_lk, deads = (int)w_dd;
  if (!mine_live) { mx = my = mz = 0.0f; }
  TE_WSTAMP(1, 1);

  // ---- P1: own slot: flags, LIDAR cell in the agent's frame (fused_lidar.py:143-217, lidar_math.py:53-83,262-311) ------------------------
  const uint32_t a_me = (mine_live && marmed_w != zero && valid) ? 1u : 0u;
  {
    const float n = fnorm(V3{mx, my, mz});
    Lor(R.accS(), a_me << s); Lor(R.accZone(), (a_me & (n > c.dome_radius ? 1u : 0u)) << s); Lor(R.accOrg(), (a_me & (n < c.origin_range ? 1u : 0u)) << s);
    L(R.pos(0, s)) = __float_as_uint(mx); L(R.pos(1, s)) = __float_as_uint(my); L(R.pos(2, s)) = __float_as_uint(mz);
  }
  int cj = 0; float rh = 1.0f;
  if (s >= 1) {
    const M3 Rm = x_inverse_attitude(ag[0], ag[1], ag[2]);
    lidar_cell_fast(c, x_mul(Rm, sub(V3{mx, my, mz}, apos)), cj, rh);
    *reinterpret_cast<uint2*>(&sm[R.cellr(s) * 64 + 2 * lane]) = make_uint2((uint32_t)cj, __float_as_uint(rh));
  }
  TE_WSTAMP(2, 0);
  TE_SLOT_BARRIER();
  TE_WSTAMP(3, 0);
  const uint32_t S = L(R.accS()), zone = L(R.accZone()), org = L(R.accOrg());
  // ---- pursuer wave: identify_closest_invader (offsets_handler.py:256-281), process_shoot_range_invaders /
  // process_explosion_range_invaders (exp03_vFinal_task.py:359-413) for ITS pursuer: none of it depends on another pursuer's outcome
  if (is_p) {
    int tgt = -1; float dmin = 0.0f;
    for (uint32_t m = live & inv_bits; m; m &= m - 1u) {   // strict '<' in slot order, over the slots somebody of the chunk has armed
      const int j = __ffs((int)m) - 1;
      const float d = fdist(V3{mx, my, mz}, V3{Lf(R.pos(0, j)), Lf(R.pos(1, j)), Lf(R.pos(2, j))});
      const bool take = a_me != 0u && ((S >> j) & 1u) != 0u && (tgt < 0 || d < dmin);
      tgt = take ? j : tgt;
      dmin = take ? d : dmin;
    }
    uint32_t rec = (uint32_t)(tgt + 1);
    if (a_me && tgt >= 0 && dmin < c.shoot_range && gun_available(c, mun, lf, step) && mun > 0) {
      mun -= 1; lf = step;
      io.st(TE_D_MUNITION, s, (uint32_t)mun); io.st(TE_D_LAST_FIRED, s, (uint32_t)step);
      const U4 r = env_rng(c, env, RNG_HIT, (uint32_t)s, 0, episode, (uint32_t)step);
      if (u01(r.x) < c.hit_prob) {  // gun.py:94; entities_manager.shoot_by_ids (:238-248)
        rec |= SLOT_HIT;
        if (c.evaluation) g.si(TE_D_KILLS, s, g.gi(TE_D_KILLS, s) + 1);  // lw_kills (evaluation_task.py:498-499)
      }
    }
    if (a_me && tgt >= 0 && dmin < c.explosion_range) rec |= SLOT_EXPLODE | (mun == 0 ? SLOT_SUICIDE : 0u);
    L(R.prec(s)) = rec;
  }
  TE_SLOT_BARRIER();   // barrier 1b: the pursuers' records

  // ---- P3: the engagement, by every wave ----------------------------------------------------------------------------------
  uint32_t killed = 0u;
  int agent_shots = 0, ally_shots = 0, exploded = 0, pursuer_suicided = 0, agent_suicided = 0;
  for (int q = 0; q < P; ++q) {
    const uint32_t r = L(R.prec(q));
    const int t = (int)(r & 0xFFu) - 1;
    if (r & SLOT_HIT) { killed |= 1u << t; if (q == 0) agent_shots += 1; else ally_shots += 1; }
    if (r & SLOT_EXPLODE) {
      killed |= (1u << q) | (1u << t);
      if ((r & SLOT_SUICIDE) && q == 0) agent_suicided += 1;
      else if (r & SLOT_SUICIDE) pursuer_suicided += 1;
      else exploded += 1;
    }
  }
  // process_invaders_in_origin (:656-659); commented out in Evaluation_Task.on_step_middle (evaluation_task.py:397)
  if ((!c.evaluation || (c.evaluation & TE_EVAL_ORIGIN_RULE)) && valid) killed |= org & inv_bits;
  const uint32_t A = S & ~killed;
  if ((killed >> s) & 1u) io.disarm(s);   // Quadcopter.disarm (quadcopter.py:461-478) of this wave's drone
  // increment_max_step (:150-153), compute_termination (:517-569)
  if (agent_shots + ally_shots > 0) max_step += c.step_increment;
  const int armed_invaders = __popc(A & inv_bits), armed_pursuers = __popc(A & pur_bits);
  const bool all_rounds_over = armed_invaders == 0 && round >= c.n_rounds;
  bool term;
  if (c.evaluation) term = (c.max_step > 0 && step > max_step) || all_rounds_over || zone != 0u || armed_pursuers == 0;
  else term = step > max_step || all_rounds_over || zone != 0u || armed_pursuers == 0 || (c.agent_death_terminates && !(A & 1u)) || apos.z < -5.99f;
  const bool to_terminal = valid && term && c.auto_reset;
  // ---- closer wins (lidar_math.py:262-311): the slot-order loop ends with, in every cell, the armed drone of smallest (range, slot);
  // a drone at range 1.0 never owns a cell (an empty cell holds 1.0).  Only slots some env of the chunk has armed can contest a cell.
  bool own = false;
  if (s >= 1) {
    own = ((A >> s) & 1u) != 0u && rh < 1.0f;
    for (uint32_t m = live & ~1u & ~(1u << s); m; m &= m - 1u) {
      const int k = __ffs((int)m) - 1;
      const uint2 cr = *reinterpret_cast<const uint2*>(&sm[R.cellr(k) * 64 + 2 * lane]);
      const float rk = __uint_as_float(cr.y);
      const bool beaten = ((A >> k) & 1u) != 0u && cr.x == (uint32_t)cj && (rk < rh || (rk == rh && k < s));
      own = own && !beaten;
    }
    if (own && valid) Lor(R.accOwn(), 1u << s);
  }
  const float flag_me = (float)(s < P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;   // lidar_math.py:305
  const bool time_plane = c.lidar_channels != 2;
  auto patch = [&](float* dst) {   // time plane = Delta 1 of a 10-deep ring (perception_snapshot.py:36-37)
    dst += (size_t)env * lidar_words(c);
    dst[cj] = rh; dst[TE_LIDAR_CELLS + cj] = flag_me;
    if (time_plane) dst[2 * TE_LIDAR_CELLS + cj] = 0.1f;
  };
  if (valid && own && !to_terminal && o.obs.lidar) patch(o.obs.lidar);
  // terminal tiles: ones, every wave its share; acknowledged before barrier 2, behind which the owners patch them
  const unsigned long long term_b = __ballot(to_terminal);
  if (o.term.lidar && term_b && is_p) {   // (the pursuer waves: an invader wave may have retired)
    for (unsigned long long tb = term_b; tb; tb &= tb - 1) {
      const int l = __ffsll((long long)tb) - 1;
      float* tile = o.term.lidar + (size_t)(blockIdx.x * 64 + l) * lidar_words(c);
      for (int e = s * 64 + lane; e < lidar_words(c); e += 64 * P) tile[e] = 1.0f;
    }
  }

  // ---- wave 0: reward, env record, outputs (exp03_vFinal_task.py:423-578) ----------------------------------------------------
  auto pos_of = [&](int t) { return V3{__uint_as_float(sm[R.pos(0, t) * 64 + lane]), __uint_as_float(sm[R.pos(1, t) * 64 + lane]), __uint_as_float(sm[R.pos(2, t) * 64 + lane])}; };
  if (s == 0) {
    if (valid) {
      io.stef(TE_E_LAST_ACTION + 0, act.x); io.stef(TE_E_LAST_ACTION + 1, act.y); io.stef(TE_E_LAST_ACTION + 2, act.z); io.stef(TE_E_LAST_ACTION + 3, act.w);
      io.ste(TE_E_STEP, (uint32_t)step);
      io.ste(TE_E_SNAP_MASK, S); io.ste(TE_E_SNAP_MASK_HI, 0u);
    }
    agent_kills += agent_shots; allies_kills += ally_shots; deads += exploded;
    if (to_terminal) {
      if (o.term.inertial) inertial_row_regs(c, o.term.inertial + (size_t)env * TE_OBS_INERTIAL_WORDS, mx, my, mz, ag, mun, lf, step);
      if (o.term.last_action) reinterpret_cast<float4*>(o.term.last_action)[env] = act;
    }
    float reward = 0.0f, cur_dist = last_dist;
    if (!c.evaluation) {
      float gs[3];
      gun_state(c, mun, lf, step, max_munition_of(c, 0), gs);
      const float dist_origin = fnorm(apos);
      int ally = -1;  // identify_closest_ally (offsets_handler.py:167-190)
      if ((S & 1u) && __popc(S & pur_bits) > 1) {
        float bd = 0.0f;
        for (int a = 1; a < P; ++a) {
          if ((S >> a) & 1u) {
            const float d = fdist(pos_of(a), apos);
            if (ally < 0 || d < bd) { ally = a; bd = d; }
          }
        }
      }
      const int chooser = ally < 0 ? 0 : ally;   // the reward's target: the closest invader of the agent's closest ally, or of the agent alone
      const int target = ((S >> chooser) & 1u) ? (int)(sm[R.prec(chooser) * 64 + lane] & 0xFFu) - 1 : -1;
      const V3 tp = target >= 0 ? pos_of(target) : V3{0.0f, 0.0f, 0.0f};
      cur_dist = fdist(apos, tp);
      const bool ready = gs[2] == 1.0f || gs[0] == 0.0f;
      float bonus = 0.0f, penalty = 0.0f;
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
    if (valid) {
      if (!c.evaluation) io.stef(TE_E_LAST_DIST, cur_dist);
      io.ste(TE_E_AGENT_KILLS, (uint32_t)agent_kills); io.ste(TE_E_ALLIES_KILLS, (uint32_t)allies_kills); io.ste(TE_E_DEADS, (uint32_t)deads);
      if (agent_shots + ally_shots > 0) io.ste(TE_E_MAX_STEP, (uint32_t)max_step);
      o.reward[env] = reward;
      o.done[env] = term ? 1 : 0;
      reinterpret_cast<int4*>(o.info)[env] = make_int4(agent_kills, allies_kills, deads, round);
      io.ste(TE_E_INFO_WAVE, (uint32_t)round);   // (on_step_end below may start the next wave; an auto-reset puts 1 back)
    }
  }

  // ---- on_step_end (:321-333): next wave when this one is cleared and a pursuer is alive; SB3 auto-reset (every wave decides, wave 0 stores)
  uint32_t task = 0u;   // round | reset << 8: the slots of this env have to be respawned
  uint32_t snap_mask = S;
  auto mask_after_spawn = [&](int rnd, bool reset) {
    const uint32_t m = reset ? pur_bits : (A & pur_bits);
    const int n = invaders_in_round(c, rnd);
    return (uint32_t)(m | ((((1u << n) - 1u) << P) & all_bits));
  };
  if (valid && !term && armed_invaders == 0 && armed_pursuers > 0) {
    round = round + (round < c.n_rounds ? 1 : c.n_rounds);  // advance_round (:155-175)
    snap_mask = mask_after_spawn(round, false);
    if (s == 0) { io.ste(TE_E_ROUND, (uint32_t)round); io.ste(TE_E_SNAP_MASK, snap_mask); io.ste(TE_E_SNAP_MASK_HI, 0u); }
    task = (uint32_t)round;
  }
  if (to_terminal) {  // Env.reset -> Task.on_reset (exp03_vFinal_environment.py:128-146): the env record by wave 0, every slot by its wave
    episode += 1u; step = 0; max_step = c.max_step; round = 1;
    snap_mask = mask_after_spawn(1, true);
    if (s == 0) {
      io.ste(TE_E_EPISODE, episode); io.ste(TE_E_STEP, 0u); io.ste(TE_E_MAX_STEP, (uint32_t)c.max_step); io.ste(TE_E_ROUND, 1u); io.ste(TE_E_INFO_WAVE, 1u);
      io.ste(TE_E_AGENT_KILLS, 0u); io.ste(TE_E_ALLIES_KILLS, 0u); io.ste(TE_E_DEADS, 0u);
      if (c.reward_model != TE_REWARD_L5_C1) io.stef(TE_E_LAST_DIST, c.dome_radius);
#pragma unroll
      for (int k = 0; k < 4; ++k) io.ste(TE_E_LAST_ACTION + k, 0u);
      io.ste(TE_E_SNAP_MASK, snap_mask); io.ste(TE_E_SNAP_MASK_HI, 0u);
    }
    act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) ag[k] = 0.0f;
    task = 1u | (1u << 8);
  }
  // ---- spawn of this wave's slot: every lane whose env starts a round or resets draws its own position (Task.setup_round / on_reset)
  if (task != 0u) {
    const bool reset = (task >> 8) != 0u;
    V3 w{0.0f, 0.0f, 0.0f};
    const bool placed = io.spawn_uniform(c, env, s, (int)(task & 0xFFu), episode, reset, ((A >> s) & 1u) != 0u, w);
    if (placed) { mx = w.x; my = w.y; mz = w.z; }
    if (reset && is_p) { mun = max_munition_of(c, s); lf = -c.cooldown_steps; fx = mx; fy = my; fz = mz; }
  }
  if (s == 0 && __ballot(task != 0u) != 0ull) {   // the retired waves' share of Task.setup_round / on_reset: their invaders go back to WaitState
    for (uint32_t m = inv_bits & ~live; m; m &= m - 1u) {
      const int k2 = __ffs((int)m) - 1;
      if (__ballot(valid && may_be_spawned(k2, (int)w_round)) != 0ull) continue;   // that wave stayed and does it itself (spawn_slot_at)
      if (task != 0u) io.st(TE_D_NAV_STATE, k2, (uint32_t)TE_NAV_WAIT);
    }
  }
  const uint32_t armed_post = task != 0u ? snap_mask : A;   // after the spawn: the pursuers as they are (all armed on reset) + the round's invaders
  L(R.npos(0, s)) = __float_as_uint(mx); L(R.npos(1, s)) = __float_as_uint(my); L(R.npos(2, s)) = __float_as_uint(mz);
  if (o.term.lidar && term_b) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the terminal tiles has landed
  TE_WSTAMP(4, 0);
  TE_SLOT_BARRIER();
  TE_WSTAMP(5, 0);

  // ---- P4 ---------------------------------------------------------------------------------------------------------------------
  if (s >= 1 && to_terminal && own && o.term.lidar) patch(o.term.lidar);
  if (o.persist && o.obs.lidar && valid) {   // persistent observation: which cells of the MAIN buffer hold a feature now, in slot order
    const uint32_t owners = L(R.accOwn());
    uint16_t* pv = o.prev + env;
    if (s == 0) pv[0] = (uint16_t)(to_terminal ? 0 : __popc(owners));
    else if (own && !to_terminal) pv[(size_t)(__popc(owners & ((1u << s) - 1u)) + 1) * p.Npad] = (uint16_t)cj;
  }
  if (s == 0) {   // the observation of the state the step leaves (post-reset values for an auto-reset env): 15 words per lane, 60 contiguous bytes
    if (o.obs.inertial && valid) inertial_row_regs(c, o.obs.inertial + (size_t)env * TE_OBS_INERTIAL_WORDS, mx, my, mz, ag, mun, lf, step);
    if (valid && o.obs.last_action) reinterpret_cast<float4*>(o.obs.last_action)[env] = act;
  }
  // ---- pursuer waves: the position the invaders steer at during the next sub-step launch (TE_X_REF), the scripted ally's command of the
  // next step (Task.on_step_start -> LoyalWingmanBehaviorTree.update, loyalwingman_navigator.py:238-352)
  if (is_p && valid) {
    int first_skipped = -1;  // drive_loyalwingmen: get_armed_pursuers()[1:] — with the agent dead the first armed ally is skipped
    if (!scripted && !(armed_post & 1u)) first_skipped = (armed_post & pur_bits & ~1u) ? __ffs((int)(armed_post & pur_bits & ~1u)) - 1 : -1;
    io.stf(TE_X_REF + 0, s, mx); io.stf(TE_X_REF + 1, s, my); io.stf(TE_X_REF + 2, s, mz);
    if ((s > 0 || scripted) && ((armed_post >> s) & 1u) && s != first_skipped) {
      float out[3] = {0.0f, 0.0f, 0.0f};
      const bool ext = driven_externally(c, s);
      if (!ext && c.ally_policy == TE_ALLY_BT) {
        const V3 me{mx, my, mz};
        if (gun_available(c, mun, lf, step)) {
          int t = -1; float bd = 0.0f; V3 tp{0.0f, 0.0f, 0.0f};
          if ((snap_mask >> s) & 1u) {
            for (int j = P; j < D; ++j) {
              if (__ballot((snap_mask >> j) & 1u) == 0ull) continue;   // nobody of the chunk has slot j armed: its row is not read
              if ((snap_mask >> j) & 1u) {
                const V3 pj{Lf(R.npos(0, j)), Lf(R.npos(1, j)), Lf(R.npos(2, j))};
                const float d = fdist(me, pj);
                if (t < 0 || d < bd) { t = j; bd = d; tp = pj; }
              }
            }
          }
          x_cmd_toward(me, tp, c.ally_speed, out);
        } else x_cmd_toward(me, V3{fx, fy, fz}, c.ally_speed, out);
      } else if (ext || c.ally_policy != TE_ALLY_FROZEN) {  // nobody / the caller's policy: the set-point persists
        out[0] = g.gf(TE_D_SETPOINT + 0, s); out[1] = g.gf(TE_D_SETPOINT + 1, s); out[2] = g.gf(TE_D_SETPOINT + 3, s);
      }
      io.stf(TE_X_CMD + 0, s, out[0]); io.stf(TE_X_CMD + 1, s, out[1]); io.stf(TE_X_CMD + 2, s, out[2]);
    }
  }
  // ---- what the next sub-step launch has to fly for this chunk (post-spawn flags): a wave walks the slots up to its own and writes its
  // items; wave 0 walks them all and writes the chunk's masks
  const int writer = 0;   // (always there; an invader wave may have retired.  The agent's wave has the shortest P4: rows only, no behaviour tree)
  if (s >= 1 || s == writer) {
    uint64_t dense = 0u, livem = 0u; int n = 0;
    uint16_t* items = p.mixed_items + (size_t)blockIdx.x * kMixedCap;
    const int k_last = s == writer ? D - 1 : s;
    for (int k = 0; k <= k_last; ++k) {
      const bool a = valid && ((armed_post >> k) & 1u) != 0u;
      const unsigned long long b = __ballot(a);
      const int cnt = __popcll(b);
      if (cnt == 0) continue;
      livem |= (uint64_t)1 << k;
      if (k == 0 || cnt >= p.dense_min || n + cnt > kMixedCap) { dense |= (uint64_t)1 << k; continue; }
      if (k == s && a) items[n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = (uint16_t)(lane | (k << 8));
      n += cnt;
    }
    if (s == writer && lane == 0) { p.slot_mask[blockIdx.x] = dense; p.mixed_count[blockIdx.x] = (uint32_t)n; p.live_mask[blockIdx.x] = livem; }
  }
  TE_WSTAMP(6, 0);
  TE_WSTAMP(7, 1);
}

// ---- several slots per wave: wave w carries the slots w, w + W, ... (SPW of them, W = ceil(D / SPW) waves), so that the P wingmen are the
// FIRST slot of the first P waves (P <= W, D <= 32).  Two uses:
//   LIDAR = false  the level5 family (cfg.stacked_obs: te_step_stacked): more drones than a workgroup has waves (level5: 6 + 12), and no own
//                  sphere — ring_push_kernel / stack_view_kernel draw every sphere from the snapshot planes this kernel leaves (SnapRows);
//   LIDAR = true   the level4 family with the own sphere, where engage_slots_kernel's one slot per wave does not fit: level5_2bt's 2 + 30
//                  drones (engage 53.5 -> 39.7 us at 65 536 envs).  On shapes both serve engage_slots_kernel stays ahead at every shard size
//                  (most of its waves retire at once; profiles/r04_q_ab_slots_per_wave.txt); TE_SLOT_SPW=2 forces this one (tests).
// Bit for bit what engage_kernel<PM, IM> leaves (and engage_slots_kernel, where both serve).  What the
// lone wave of engage_kernel spends at 65 536 envs (tools/engage_stamps.py, level5): requests 6.5 us, closest invaders 3.5, shots 3.8,
// snapshot planes + reward 7, rows 2.5, the five allies' behaviour trees 9, plan 1.3 — here every one of these is one slot's share.
// LDS rows of engage_slots_multi_kernel (256 B each): one record row per wingman; the {cell, range} pairs and the owner row behind the rest
// (LIDAR only) — 30 KB for level5, four workgroups per CU
struct MultiSlotRows {
  int D, P, H;   // H = 32-bit halves of a slot mask: 1, or 2 beyond 32 drones (level5_fusion / level5_dumb)
  TE_DEV int accS(int h = 0) const { return h; }
  TE_DEV int accZone(int h = 0) const { return H + h; }
  TE_DEV int accOrg(int h = 0) const { return 2 * H + h; }
  TE_DEV int pos(int k, int s) const { return 3 * H + k * D + s; }
  TE_DEV int npos(int k, int s) const { return 3 * H + (3 + k) * D + s; }
  TE_DEV int prec(int q) const { return 3 * H + 6 * D + q; }
  TE_DEV int accOwn() const { return 3 * H + 6 * D + P; }
  TE_DEV int cellr(int s) const { return 3 * H + 1 + 6 * D + P + 2 * s; }   // 8-byte {cell, range} per lane: two rows
};
__host__ __device__ inline int multi_slot_lds_rows(int D, int P, bool lidar) { return 3 * (D > 32 ? 2 : 1) + 6 * D + P + (lidar ? 1 + 2 * D : 0); }

template <int SPW, bool LIDAR, bool WIDE = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(SPW == 1 ? 4 : 6))) void engage_slots_multi_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  TE_EXACT
  extern __shared__ uint32_t sm[];
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers;
  const int W = (int)(blockDim.x >> 6);
  using M = typename std::conditional<WIDE, uint64_t, uint32_t>::type;   // one bit per slot (WIDE: up to 39 drones in 13 waves of 3)
  constexpr M one = 1;
  constexpr int kSh = WIDE ? 63 : 31;
  static_assert(!(WIDE && LIDAR), "the own sphere's owner row is 32 bits");
  const MultiSlotRows R{D, P, WIDE ? 2 : 1};
  auto popcM = [](M m) { return WIDE ? __popcll((unsigned long long)m) : __popc((uint32_t)m); };
  auto ffsM = [](M m) { return WIDE ? __ffsll((long long)m) - 1 : __ffs((int)(uint32_t)m) - 1; };
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // this wave; its first slot
  const int env = blockIdx.x * 64 + lane;
  const bool valid = env < p.N;
  const GView g{p.dstate, p.estate, D, p.Npad, env, P};
  const EnvIO io(p, env);
  const M pur_bits = (one << P) - one, all_bits = D >= (int)(8 * sizeof(M)) ? ~(M)0 : (one << D) - one, inv_bits = all_bits & ~pur_bits;
  const bool scripted = all_scripted(c);
  const bool is_p = s < P;
  int sl[SPW]; bool has[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) { sl[u] = s + u * W; has[u] = sl[u] < D; }
  auto L = [&](int row) -> uint32_t& { return sm[row * 64 + lane]; };
  auto Lf = [&](int row) { return __uint_as_float(sm[row * 64 + lane]); };
  auto Lor = [&](int row, uint32_t v) { __hip_atomic_fetch_or(&sm[row * 64 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  TE_WSTAMP(0, 0);

  // ---- P0 (see engage_slots_kernel) -------------------------------------------------------------------------------------------------------
  const uint32_t* __restrict__ lm32 = reinterpret_cast<const uint32_t*>(p.live_mask);
  const M live = (WIDE ? (M)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(lm32[2 * blockIdx.x]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(lm32[2 * blockIdx.x + 1]) << 32))
                       : (M)(uint32_t)__builtin_amdgcn_readfirstlane(lm32[2 * blockIdx.x])) & all_bits;
  auto next_round_of = [&](int r) { return r + (r < c.n_rounds ? 1 : c.n_rounds); };   // advance_round (exp03_vFinal_task.py:155-175)
  auto may_be_spawned = [&](int slot, int r) { const int i = slot - P; return i < invaders_in_round(c, next_round_of(r)) || i < invaders_in_round(c, 1); };
  bool mine_live[SPW]; bool any_live = is_p;
#pragma unroll
  for (int u = 0; u < SPW; ++u) { mine_live[u] = has[u] && ((u == 0 && is_p) || ((live >> (sl[u] & kSh)) & one) != 0); any_live = any_live || mine_live[u]; }
  if (!any_live) {   // none of this wave's slots is armed anywhere in the chunk, none can be armed by this step: the wave retires
    const int r = (int)io.le(TE_E_ROUND);
    bool maybe = false;
#pragma unroll
    for (int u = 0; u < SPW; ++u) maybe = maybe || (has[u] && may_be_spawned(sl[u], r));
    if (__ballot(valid && maybe) == 0ull) return;
  }
  float mx[SPW], my[SPW], mz[SPW]; uint32_t marmed_w[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) {   // (a slot beyond D repeats the first one's rows: every request unconditional)
    const int ls = has[u] ? sl[u] : s;
    mx[u] = io.ldf(TE_D_OBS_POS, ls); my[u] = io.ldf(TE_D_OBS_POS + 1, ls); mz[u] = io.ldf(TE_D_OBS_POS + 2, ls);
    marmed_w[u] = io.ld(TE_D_ARMED, ls);
  }
  const uint32_t w_step = io.le(TE_E_STEP), w_max_step = io.le(TE_E_MAX_STEP), w_round = io.le(TE_E_ROUND), w_episode = io.le(TE_E_EPISODE);
  float ag[9];  // OBS_EULER, OBS_VEL, OBS_RATE of the agent: wave 0 (reward, rows); with the own sphere every wave needs the attitude and the position
#pragma unroll
  for (int k = 0; k < 9; ++k) ag[k] = 0.0f;
  V3 apos0{0.0f, 0.0f, 0.0f};
  if (LIDAR) {
#pragma unroll
    for (int k = 0; k < 3; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
    apos0 = V3{io.ldf(TE_D_OBS_POS, 0), io.ldf(TE_D_OBS_POS + 1, 0), io.ldf(TE_D_OBS_POS + 2, 0)};
  }
  uint32_t w_last_dist = 0u, w_ak = 0u, w_lk = 0u, w_dd = 0u;
  float4 act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (s == 0) {
#pragma unroll
    for (int k = LIDAR ? 3 : 0; k < 9; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
    w_last_dist = io.le(TE_E_LAST_DIST); w_ak = io.le(TE_E_AGENT_KILLS); w_lk = io.le(TE_E_ALLIES_KILLS); w_dd = io.le(TE_E_DEADS);
    if (valid) act = reinterpret_cast<const float4*>(actions)[env];
  }
  uint32_t w_mun = 0u, w_lf = 0u, eul[3] = {0u, 0u, 0u}; float fx = 0.0f, fy = 0.0f, fz = 0.0f;
  if (is_p) {   // gun, formation point, and the attitude the snapshot planes keep of a wingman
    w_mun = io.ld(TE_D_MUNITION, s); w_lf = io.ld(TE_D_LAST_FIRED, s);
    fx = io.ldf(TE_D_FORMATION, s); fy = io.ldf(TE_D_FORMATION + 1, s); fz = io.ldf(TE_D_FORMATION + 2, s);
#pragma unroll
    for (int k = 0; k < 3; ++k) eul[k] = io.ld(TE_D_OBS_EULER + k, s);
  }
  uint32_t zero = 0u;
  asm volatile("" : "+s"(zero));
  if (s == 0) {
#pragma unroll
    for (int h = 0; h < (WIDE ? 2 : 1); ++h) { L(R.accS(h)) = 0u; L(R.accZone(h)) = 0u; L(R.accOrg(h)) = 0u; }
    if (LIDAR) L(R.accOwn()) = 0u;
  }
  TE_SLOT_BARRIER();   // barrier 0: the accumulator rows are zero
  int step = (int)w_step + 1;  // AGENT_STEP_BROADCAST (exp03_vFinal_environment.py:177-182)
  int max_step = (int)w_max_step, round = (int)w_round;
  uint32_t episode = w_episode;
  int mun = (int)w_mun, lf = (int)w_lf;
  const float last_dist = __uint_as_float(w_last_dist);
  int agent_kills = (int)w_ak, allies_kills = (int)w_lk, deads = (int)w_dd;
  TE_WSTAMP(1, 1);

  // ---- P1: own slots: flags and positions -------------------------------------------------------------------------------------------------
  uint32_t a_me[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    if (!mine_live[u]) { mx[u] = my[u] = mz[u] = 0.0f; }
    a_me[u] = (mine_live[u] && marmed_w[u] != zero && valid) ? 1u : 0u;
    if (has[u]) {
      const float n = fnorm(V3{mx[u], my[u], mz[u]});
      const int sh = sl[u] & 31, hh = WIDE ? (sl[u] >> 5) & 1 : 0;   // the half that holds this slot's bit
      Lor(R.accS(hh), a_me[u] << sh); Lor(R.accZone(hh), (a_me[u] & (n > c.dome_radius ? 1u : 0u)) << sh); Lor(R.accOrg(hh), (a_me[u] & (n < c.origin_range ? 1u : 0u)) << sh);
      L(R.pos(0, sl[u])) = __float_as_uint(mx[u]); L(R.pos(1, sl[u])) = __float_as_uint(my[u]); L(R.pos(2, sl[u])) = __float_as_uint(mz[u]);
    }
  }
  // own slots' LIDAR cells in the agent's frame (fused_lidar.py:143-217, lidar_math.py:53-83,262-311)
  int cj[SPW]; float rh[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) { cj[u] = 0; rh[u] = 1.0f; }
  if (LIDAR) {
    const M3 Rm = x_inverse_attitude(ag[0], ag[1], ag[2]);
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      if (has[u] && sl[u] >= 1) {
        lidar_cell_fast(c, x_mul(Rm, sub(V3{mx[u], my[u], mz[u]}, apos0)), cj[u], rh[u]);
        *reinterpret_cast<uint2*>(&sm[R.cellr(sl[u]) * 64 + 2 * lane]) = make_uint2((uint32_t)cj[u], __float_as_uint(rh[u]));
      }
    }
  }
  TE_WSTAMP(2, 0);
  TE_SLOT_BARRIER();
  TE_WSTAMP(3, 0);
  auto rdM = [&](int lo, int hi) { return WIDE ? (M)((uint64_t)L(lo) | ((uint64_t)L(hi) << 32)) : (M)L(lo); };
  const M S = rdM(R.accS(0), R.accS(1)), zone = rdM(R.accZone(0), R.accZone(1)), org = rdM(R.accOrg(0), R.accOrg(1));
  const V3 apos{Lf(R.pos(0, 0)), Lf(R.pos(1, 0)), Lf(R.pos(2, 0))};
  // ---- pursuer wave: identify_closest_invader (offsets_handler.py:256-281), process_shoot_range_invaders /
  // process_explosion_range_invaders (exp03_vFinal_task.py:359-413) for ITS pursuer
  if (is_p) {
    int tgt = -1; float dmin = 0.0f;
    for (M m = live & inv_bits; m; m &= m - one) {   // strict '<' in slot order, over the slots somebody of the chunk has armed
      const int j = ffsM(m);
      const float d = fdist(V3{mx[0], my[0], mz[0]}, V3{Lf(R.pos(0, j)), Lf(R.pos(1, j)), Lf(R.pos(2, j))});
      const bool take = a_me[0] != 0u && ((S >> j) & one) != 0 && (tgt < 0 || d < dmin);
      tgt = take ? j : tgt;
      dmin = take ? d : dmin;
    }
    uint32_t rec = (uint32_t)(tgt + 1);
    if (a_me[0] && tgt >= 0 && dmin < c.shoot_range && gun_available(c, mun, lf, step) && mun > 0) {
      mun -= 1; lf = step;
      io.st(TE_D_MUNITION, s, (uint32_t)mun); io.st(TE_D_LAST_FIRED, s, (uint32_t)step);
      const U4 r = env_rng(c, env, RNG_HIT, (uint32_t)s, 0, episode, (uint32_t)step);
      if (u01(r.x) < c.hit_prob) {  // gun.py:94; entities_manager.shoot_by_ids (:238-248)
        rec |= SLOT_HIT;
        if (c.evaluation) g.si(TE_D_KILLS, s, g.gi(TE_D_KILLS, s) + 1);  // lw_kills (evaluation_task.py:498-499)
      }
    }
    if (a_me[0] && tgt >= 0 && dmin < c.explosion_range) rec |= SLOT_EXPLODE | (mun == 0 ? SLOT_SUICIDE : 0u);
    L(R.prec(s)) = rec;
  }
  TE_SLOT_BARRIER();   // barrier 1b: the pursuers' records

  // ---- P3: the engagement, by every wave ----------------------------------------------------------------------------------------------------
  M killed = 0;
  int agent_shots = 0, ally_shots = 0, exploded = 0, pursuer_suicided = 0, agent_suicided = 0;
  for (int q = 0; q < P; ++q) {
    const uint32_t r = L(R.prec(q));
    const int t = (int)(r & 0xFFu) - 1;
    if (r & SLOT_HIT) { killed |= one << t; if (q == 0) agent_shots += 1; else ally_shots += 1; }
    if (r & SLOT_EXPLODE) {
      killed |= (one << q) | (one << t);
      if ((r & SLOT_SUICIDE) && q == 0) agent_suicided += 1;
      else if (r & SLOT_SUICIDE) pursuer_suicided += 1;
      else exploded += 1;
    }
  }
  // process_invaders_in_origin (:656-659); commented out in Evaluation_Task.on_step_middle (evaluation_task.py:397)
  if ((!c.evaluation || (c.evaluation & TE_EVAL_ORIGIN_RULE)) && valid) killed |= org & inv_bits;
  const M A = S & ~killed;
#pragma unroll
  for (int u = 0; u < SPW; ++u)
    if (has[u] && ((killed >> (sl[u] & kSh)) & one)) io.disarm(sl[u]);   // Quadcopter.disarm (quadcopter.py:461-478)
  // increment_max_step (:150-153), compute_termination (:517-569)
  if (agent_shots + ally_shots > 0) max_step += c.step_increment;
  const int armed_invaders = popcM(A & inv_bits), armed_pursuers = popcM(A & pur_bits);
  const bool all_rounds_over = armed_invaders == 0 && round >= c.n_rounds;
  bool term;
  if (c.evaluation) term = (c.max_step > 0 && step > max_step) || all_rounds_over || zone != 0u || armed_pursuers == 0;
  else term = step > max_step || all_rounds_over || zone != 0u || armed_pursuers == 0 || (c.agent_death_terminates && !(A & one)) || apos.z < -5.99f;
  const bool to_terminal = valid && term && c.auto_reset;
  // ---- closer wins (lidar_math.py:262-311) as a fixed point: in every cell the armed drone of smallest (range, slot); a drone at range 1.0
  // never owns a cell.  Only slots some env of the chunk has armed can contest a cell.  Patches: flag = type / 5 (lidar_math.py:305), the
  // time plane is Delta 1 of a 10-deep ring (perception_snapshot.py:36-37)
  bool own[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) own[u] = false;
  const bool time_plane = c.lidar_channels != 2;
  auto patch = [&](float* dst, int cell, float range, int slot) {
    dst += (size_t)env * lidar_words(c);
    dst[cell] = range; dst[TE_LIDAR_CELLS + cell] = (float)(slot < P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;
    if (time_plane) dst[2 * TE_LIDAR_CELLS + cell] = 0.1f;
  };
  unsigned long long term_b = 0ull;
  if constexpr (LIDAR) {
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      if (has[u] && sl[u] >= 1) {
        const int me = sl[u];
        bool mine = ((A >> (me & 31)) & 1u) != 0u && rh[u] < 1.0f;
        for (uint32_t m = live & ~1u & ~(1u << (me & 31)); m; m &= m - 1u) {
          const int k2 = __ffs((int)m) - 1;
          const uint2 cr = *reinterpret_cast<const uint2*>(&sm[R.cellr(k2) * 64 + 2 * lane]);
          const float rk = __uint_as_float(cr.y);
          const bool beaten = ((A >> k2) & 1u) != 0u && cr.x == (uint32_t)cj[u] && (rk < rh[u] || (rk == rh[u] && k2 < me));
          mine = mine && !beaten;
        }
        own[u] = mine;
        if (mine && valid) Lor(R.accOwn(), 1u << (me & 31));
        if (valid && mine && !to_terminal && o.obs.lidar) patch(o.obs.lidar, cj[u], rh[u], me);
      }
    }
    // terminal tiles: ones, the pursuer waves their shares (an invader wave may have retired); acknowledged before barrier 2, behind which the owners patch them
    term_b = __ballot(to_terminal);
    if (o.term.lidar && term_b && is_p) {
      for (unsigned long long tb = term_b; tb; tb &= tb - 1) {
        const int l = __ffsll((long long)tb) - 1;
        float* tile = o.term.lidar + (size_t)(blockIdx.x * 64 + l) * lidar_words(c);
        for (int e = s * 64 + lane; e < lidar_words(c); e += 64 * P) tile[e] = 1.0f;
      }
    }
  }
  // ---- what this step's stacked observation may look at, BEFORE anything respawns (te_stacked.hpp SnapRows): every wave its slots' rows
  if (p.snap && valid) {
    const SnapRows sr{D, P};
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      if (has[u]) {
        p.snap[(size_t)(sr.pos() + 0 * D + sl[u]) * p.Npad + env] = __float_as_uint(mx[u]);
        p.snap[(size_t)(sr.pos() + 1 * D + sl[u]) * p.Npad + env] = __float_as_uint(my[u]);
        p.snap[(size_t)(sr.pos() + 2 * D + sl[u]) * p.Npad + env] = __float_as_uint(mz[u]);
      }
    }
    if (is_p) {
#pragma unroll
      for (int k = 0; k < 3; ++k) p.snap[(size_t)(sr.euler() + k * P + s) * p.Npad + env] = eul[k];
    }
    if (s == 0) {
      p.snap[(size_t)sr.armed() * p.Npad + env] = (uint32_t)A; p.snap[(size_t)sr.armed_hi() * p.Npad + env] = (uint32_t)((uint64_t)A >> 32);
      p.snap[(size_t)sr.step() * p.Npad + env] = (uint32_t)step;
      p.snap[(size_t)sr.episode() * p.Npad + env] = episode;
      p.snap[(size_t)sr.done() * p.Npad + env] = to_terminal ? 1u : 0u;
    }
  }

  // ---- wave 0: reward, env record, outputs (exp03_vFinal_task.py:423-578 and the level5 reward models) -----------------------------------
  auto pos_of = [&](int t) { return V3{__uint_as_float(sm[R.pos(0, t) * 64 + lane]), __uint_as_float(sm[R.pos(1, t) * 64 + lane]), __uint_as_float(sm[R.pos(2, t) * 64 + lane])}; };
  if (s == 0) {
    if (valid) {
      io.stef(TE_E_LAST_ACTION + 0, act.x); io.stef(TE_E_LAST_ACTION + 1, act.y); io.stef(TE_E_LAST_ACTION + 2, act.z); io.stef(TE_E_LAST_ACTION + 3, act.w);
      io.ste(TE_E_STEP, (uint32_t)step);
      io.ste(TE_E_SNAP_MASK, (uint32_t)S); io.ste(TE_E_SNAP_MASK_HI, (uint32_t)((uint64_t)S >> 32));
    }
    agent_kills += agent_shots; allies_kills += ally_shots; deads += exploded;
    if (to_terminal) {
      if (o.term.inertial) inertial_row_regs(c, o.term.inertial + (size_t)env * TE_OBS_INERTIAL_WORDS, mx[0], my[0], mz[0], ag, mun, lf, step);
      if (o.term.last_action) reinterpret_cast<float4*>(o.term.last_action)[env] = act;
    }
    float reward = 0.0f, cur_dist = last_dist;
    if (!c.evaluation) {
      float gs[3];
      gun_state(c, mun, lf, step, max_munition_of(c, 0), gs);
      const float dist_origin = fnorm(apos);
      int ally = -1;  // identify_closest_ally (offsets_handler.py:167-190)
      if ((S & one) && popcM(S & pur_bits) > 1) {
        float bd = 0.0f;
        for (int a = 1; a < P; ++a) {
          if ((S >> a) & one) {
            const float d = fdist(pos_of(a), apos);
            if (ally < 0 || d < bd) { ally = a; bd = d; }
          }
        }
      }
      const int chooser = ally < 0 ? 0 : ally;   // the reward's target: the closest invader of the agent's closest ally, or of the agent alone
      const int target = ((S >> chooser) & one) ? (int)(sm[R.prec(chooser) * 64 + lane] & 0xFFu) - 1 : -1;
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
        const int t1 = (S & one) ? (int)(sm[R.prec(0) * 64 + lane] & 0xFFu) - 1 : -1;   // the agent's OWN closest invader (:458)
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
    if (valid) {
      if (!c.evaluation) io.stef(TE_E_LAST_DIST, cur_dist);
      io.ste(TE_E_AGENT_KILLS, (uint32_t)agent_kills); io.ste(TE_E_ALLIES_KILLS, (uint32_t)allies_kills); io.ste(TE_E_DEADS, (uint32_t)deads);
      if (agent_shots + ally_shots > 0) io.ste(TE_E_MAX_STEP, (uint32_t)max_step);
      o.reward[env] = reward;
      o.done[env] = term ? 1 : 0;
      reinterpret_cast<int4*>(o.info)[env] = make_int4(agent_kills, allies_kills, deads, round);
      io.ste(TE_E_INFO_WAVE, (uint32_t)round);   // (on_step_end below may start the next wave; an auto-reset puts 1 back)
    }
  }

  // ---- on_step_end (:321-333): next wave when this one is cleared and a pursuer is alive; SB3 auto-reset (every wave decides, wave 0 stores)
  uint32_t task = 0u;   // round | reset << 8: the slots of this env have to be respawned
  M snap_mask = S;
  auto mask_after_spawn = [&](int rnd, bool reset) {
    const M m = reset ? pur_bits : (A & pur_bits);
    const int n = invaders_in_round(c, rnd);
    return (M)(m | ((M)((((uint64_t)1 << n) - 1u) << P) & all_bits));
  };
  if (valid && !term && armed_invaders == 0 && armed_pursuers > 0) {
    round = round + (round < c.n_rounds ? 1 : c.n_rounds);  // advance_round (:155-175)
    snap_mask = mask_after_spawn(round, false);
    if (s == 0) { io.ste(TE_E_ROUND, (uint32_t)round); io.ste(TE_E_SNAP_MASK, (uint32_t)snap_mask); io.ste(TE_E_SNAP_MASK_HI, (uint32_t)((uint64_t)snap_mask >> 32)); }
    task = (uint32_t)round;
  }
  if (to_terminal) {  // Env.reset -> Task.on_reset (exp03_vFinal_environment.py:128-146): the env record by wave 0, every slot by its wave
    episode += 1u; step = 0; max_step = c.max_step; round = 1;
    snap_mask = mask_after_spawn(1, true);
    if (s == 0) {
      io.ste(TE_E_EPISODE, episode); io.ste(TE_E_STEP, 0u); io.ste(TE_E_MAX_STEP, (uint32_t)c.max_step); io.ste(TE_E_ROUND, 1u); io.ste(TE_E_INFO_WAVE, 1u);
      io.ste(TE_E_AGENT_KILLS, 0u); io.ste(TE_E_ALLIES_KILLS, 0u); io.ste(TE_E_DEADS, 0u);
      if (c.reward_model != TE_REWARD_L5_C1) io.stef(TE_E_LAST_DIST, c.dome_radius);
#pragma unroll
      for (int k = 0; k < 4; ++k) io.ste(TE_E_LAST_ACTION + k, 0u);
      io.ste(TE_E_SNAP_MASK, (uint32_t)snap_mask); io.ste(TE_E_SNAP_MASK_HI, (uint32_t)((uint64_t)snap_mask >> 32));
    }
    act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) ag[k] = 0.0f;
    task = 1u | (1u << 8);
  }
  // ---- spawn of this wave's slots: every lane whose env starts a round or resets draws its own positions (Task.setup_round / on_reset)
  if (task != 0u) {
    const bool reset = (task >> 8) != 0u;
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      if (has[u]) {
        V3 w{0.0f, 0.0f, 0.0f};
        const bool placed = io.spawn_uniform(c, env, sl[u], (int)(task & 0xFFu), episode, reset, ((A >> (sl[u] & kSh)) & one) != 0, w);
        if (placed) { mx[u] = w.x; my[u] = w.y; mz[u] = w.z; }
      }
    }
    if (reset && is_p) { mun = max_munition_of(c, s); lf = -c.cooldown_steps; fx = mx[0]; fy = my[0]; fz = mz[0]; }
  }
  if (s == 0 && __ballot(task != 0u) != 0ull) {   // the retired waves' share of Task.setup_round / on_reset: their invaders go back to WaitState
    for (M m = inv_bits & ~live; m; m &= m - one) {
      const int k2 = ffsM(m);
      const int wv = k2 % W;
      bool stayed = wv < P;   // did the wave that carries slot k2 stay?  (then it does this itself: spawn_slot_at)
      for (int u2 = 0; u2 < SPW && !stayed; ++u2) {
        const int s2 = wv + u2 * W;
        if (s2 < D) stayed = ((live >> s2) & one) != 0 || __ballot(valid && may_be_spawned(s2, (int)w_round)) != 0ull;
      }
      if (stayed) continue;
      if (task != 0u) io.st(TE_D_NAV_STATE, k2, (uint32_t)TE_NAV_WAIT);
    }
  }
  const M armed_post = task != 0u ? snap_mask : A;   // after the spawn: the pursuers as they are (all armed on reset) + the round's invaders
#pragma unroll
  for (int u = 0; u < SPW; ++u)
    if (has[u]) { L(R.npos(0, sl[u])) = __float_as_uint(mx[u]); L(R.npos(1, sl[u])) = __float_as_uint(my[u]); L(R.npos(2, sl[u])) = __float_as_uint(mz[u]); }
  if (LIDAR && o.term.lidar && term_b) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the terminal tiles has landed
  TE_WSTAMP(4, 0);
  TE_SLOT_BARRIER();
  TE_WSTAMP(5, 0);

  // ---- P4 --------------------------------------------------------------------------------------------------------------------------------
  if constexpr (LIDAR) {
    const uint32_t owners = L(R.accOwn());
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      if (has[u] && sl[u] >= 1) {
        if (to_terminal && own[u] && o.term.lidar) patch(o.term.lidar, cj[u], rh[u], sl[u]);
        // persistent observation: which cells of the MAIN buffer hold a feature now, in slot order
        if (o.persist && o.obs.lidar && valid && own[u] && !to_terminal)
          o.prev[env + (size_t)(__popc(owners & ((1u << (sl[u] & 31)) - 1u)) + 1) * p.Npad] = (uint16_t)cj[u];
      }
    }
    if (s == 0 && o.persist && o.obs.lidar && valid) o.prev[env] = (uint16_t)(to_terminal ? 0 : __popc(owners));
  }
  if (s == 0) {   // the observation of the state the step leaves (post-reset values for an auto-reset env)
    if (o.obs.inertial && valid) inertial_row_regs(c, o.obs.inertial + (size_t)env * TE_OBS_INERTIAL_WORDS, mx[0], my[0], mz[0], ag, mun, lf, step);
    if (valid && o.obs.last_action) reinterpret_cast<float4*>(o.obs.last_action)[env] = act;
  }
  // ---- pursuer waves: TE_X_REF and the scripted wingman's command of the next step (loyalwingman_navigator.py:238-352)
  if (is_p && valid) {
    int first_skipped = -1;  // drive_loyalwingmen: get_armed_pursuers()[1:] — with the agent dead the first armed ally is skipped
    if (!scripted && !(armed_post & one)) first_skipped = (armed_post & pur_bits & ~one) ? ffsM(armed_post & pur_bits & ~one) : -1;
    io.stf(TE_X_REF + 0, s, mx[0]); io.stf(TE_X_REF + 1, s, my[0]); io.stf(TE_X_REF + 2, s, mz[0]);
    if ((s > 0 || scripted) && ((armed_post >> s) & one) && s != first_skipped) {
      float out[3] = {0.0f, 0.0f, 0.0f};
      const bool ext = driven_externally(c, s);
      if (!ext && c.ally_policy == TE_ALLY_BT) {
        const V3 me{mx[0], my[0], mz[0]};
        if (gun_available(c, mun, lf, step)) {
          int t = -1; float bd = 0.0f; V3 tp{0.0f, 0.0f, 0.0f};
          if ((snap_mask >> s) & one) {
            for (int j = P; j < D; ++j) {
              if (__ballot(((snap_mask >> j) & one) != 0) == 0ull) continue;   // nobody of the chunk has slot j armed: its row is not read
              if ((snap_mask >> j) & one) {
                const V3 pj{Lf(R.npos(0, j)), Lf(R.npos(1, j)), Lf(R.npos(2, j))};
                const float d = fdist(me, pj);
                if (t < 0 || d < bd) { t = j; bd = d; tp = pj; }
              }
            }
          }
          x_cmd_toward(me, tp, c.ally_speed, out);
        } else x_cmd_toward(me, V3{fx, fy, fz}, c.ally_speed, out);
      } else if (ext || c.ally_policy != TE_ALLY_FROZEN) {  // nobody / the caller's policy: the set-point persists
        out[0] = g.gf(TE_D_SETPOINT + 0, s); out[1] = g.gf(TE_D_SETPOINT + 1, s); out[2] = g.gf(TE_D_SETPOINT + 3, s);
      }
      io.stf(TE_X_CMD + 0, s, out[0]); io.stf(TE_X_CMD + 1, s, out[1]); io.stf(TE_X_CMD + 2, s, out[2]);
    }
  }
  // ---- the next sub-step launch's flight plan: a wave walks the slots up to its highest one and writes its own items; wave 0
  // walks them all and writes the chunk's masks
  {
    const int writer = 0;   // (the agent's wave: no behaviour tree in its P4)
    int top = s;
#pragma unroll
    for (int u = 1; u < SPW; ++u) if (has[u]) top = sl[u];
    uint64_t dense = 0u, livem = 0u; int n = 0;
    uint16_t* items = p.mixed_items + (size_t)blockIdx.x * kMixedCap;
    const int k_last = s == writer ? D - 1 : top;
    for (int k = 0; k <= k_last; ++k) {
      const bool a = valid && ((armed_post >> k) & one) != 0;
      const unsigned long long b = __ballot(a);
      const int cnt = __popcll(b);
      if (cnt == 0) continue;
      livem |= (uint64_t)1 << k;
      if (k == 0 || cnt >= p.dense_min || n + cnt > kMixedCap) { dense |= (uint64_t)1 << k; continue; }
      bool own = false;
#pragma unroll
      for (int u = 0; u < SPW; ++u) own = own || (has[u] && k == sl[u]);
      if (own && a) items[n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = (uint16_t)(lane | (k << 8));
      n += cnt;
    }
    if (s == writer && lane == 0) { p.slot_mask[blockIdx.x] = dense; p.mixed_count[blockIdx.x] = (uint32_t)n; p.live_mask[blockIdx.x] = livem; }
  }
  TE_WSTAMP(6, 0);
  TE_WSTAMP(7, 1);
}

// ---- stage02 (L3Stage1.on_step_middle / on_step_end, level3/components/stages.py:144-179,241-344) in the same form: bit for bit what
// engage_stage02_kernel<2, 8> writes.  Differences from the level4 family: the suicide rule (a pursuer without munition kills by contact at
// shoot range), every kill counts for the agent, killed invaders come back inside the step, no waves, no behaviour tree.
enum : uint32_t { SLOT2_KILL = 1u << 8 };   // pursuer record: target + 1 | KILL (shot that hit, or the suicide rule) | SLOT_EXPLODE; its row pair holds dmin too

template <int DM>
__global__ __launch_bounds__(DM * 64) void engage_slots_stage02_kernel(Params p, const float* __restrict__ actions, StepOut o) {
  TE_EXACT
  extern __shared__ uint32_t sm[];
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers;
  const SlotRows R{D, P};
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int env = blockIdx.x * 64 + lane;
  const bool valid = env < p.N;
  const EnvIO io(p, env);
  const uint32_t pur_bits = (1u << P) - 1u, all_bits = (1u << D) - 1u, inv_bits = all_bits & ~pur_bits;
  const bool is_p = s < P;
  auto L = [&](int row) -> uint32_t& { return sm[row * 64 + lane]; };
  auto Lf = [&](int row) { return __uint_as_float(sm[row * 64 + lane]); };
  auto Lor = [&](int row, uint32_t v) { __hip_atomic_fetch_or(&sm[row * 64 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

  // ---- P0: loads
  float mx = io.ldf(TE_D_OBS_POS, s), my = io.ldf(TE_D_OBS_POS + 1, s), mz = io.ldf(TE_D_OBS_POS + 2, s);
  const uint32_t marmed_w = io.ld(TE_D_ARMED, s);
  const V3 apos{io.ldf(TE_D_OBS_POS, 0), io.ldf(TE_D_OBS_POS + 1, 0), io.ldf(TE_D_OBS_POS + 2, 0)};
  float ag[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
#pragma unroll
  for (int k = 3; k < 9; ++k) ag[k] = 0.0f;
  const uint32_t w_step = io.le(TE_E_STEP), w_max_step = io.le(TE_E_MAX_STEP), w_episode = io.le(TE_E_EPISODE);
  uint32_t w_kills = 0u, w_deads = 0u, w_last = 0u;
  float4 act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (s == 0) {
#pragma unroll
    for (int k = 3; k < 9; ++k) ag[k] = io.ldf(TE_D_OBS_EULER + k, 0);
    w_kills = io.le(TE_E_AGENT_KILLS); w_deads = io.le(TE_E_DEADS); w_last = io.le(TE_E_PREV_SNAP_MIN);
    if (valid) act = reinterpret_cast<const float4*>(actions)[env];
  }
  uint32_t w_mun = 0u, w_lf = 0u;
  float qx[DM], qy[DM], qz[DM]; uint32_t qa[DM];
#pragma unroll
  for (int j = 0; j < DM; ++j) { qx[j] = qy[j] = qz[j] = 0.0f; qa[j] = 0u; }
  if (is_p) {
    w_mun = io.ld(TE_D_MUNITION, s); w_lf = io.ld(TE_D_LAST_FIRED, s);
#pragma unroll
    for (int j = 1; j < DM; ++j) {
      if ((inv_bits >> j) & 1u) {
        qx[j] = io.ldf(TE_D_OBS_POS, j); qy[j] = io.ldf(TE_D_OBS_POS + 1, j); qz[j] = io.ldf(TE_D_OBS_POS + 2, j); qa[j] = io.ld(TE_D_ARMED, j);
      }
    }
  }
  uint32_t zero = 0u;   // (see engage_slots_kernel)
  asm volatile("" : "+s"(zero));
  if (s == 0) { L(R.accS()) = 0u; L(R.accZone()) = 0u; L(R.accOrg()) = 0u; L(R.accOwn()) = 0u; }
  TE_SLOT_BARRIER();
  int step = (int)w_step + 1;
  const int max_step = (int)w_max_step;
  uint32_t episode = w_episode;
  int mun = (int)w_mun, lf = (int)w_lf;
  int kills = (int)w_kills, deads = (int)w_deads;
  const float last = __uint_as_float(w_last);

  // ---- P1: own slot
  const uint32_t a_me = (marmed_w != zero && valid) ? 1u : 0u;
  {
    const float n = fnorm(V3{mx, my, mz});
    Lor(R.accS(), a_me << s); Lor(R.accZone(), (a_me & (n > c.dome_radius ? 1u : 0u)) << s);
    L(R.pos(0, s)) = __float_as_uint(mx); L(R.pos(1, s)) = __float_as_uint(my); L(R.pos(2, s)) = __float_as_uint(mz);
  }
  int cj = 0; float rh = 1.0f;
  if (s >= 1) {
    const M3 Rm = x_inverse_attitude(ag[0], ag[1], ag[2]);
    lidar_cell_fast(c, x_mul(Rm, sub(V3{mx, my, mz}, apos)), cj, rh);
    *reinterpret_cast<uint2*>(&sm[R.cellr(s) * 64 + 2 * lane]) = make_uint2((uint32_t)cj, __float_as_uint(rh));
  }
  if (is_p) {   // closest invader, shoot_by_ids with the suicide rule (level3/components/quadcopter_manager.py:155-171), explosion
    int tgt = -1; float dmin = 0.0f;
#pragma unroll
    for (int j = 1; j < DM; ++j) {
      const float d = fdist(V3{mx, my, mz}, V3{qx[j], qy[j], qz[j]});
      const bool take = a_me != 0u && qa[j] != zero && (tgt < 0 || d < dmin);
      tgt = take ? j : tgt; dmin = take ? d : dmin;
    }
    uint32_t rec = (uint32_t)(tgt + 1);
    if (a_me && tgt >= 0 && dmin < c.shoot_range) {
      if (mun == 0) rec |= SLOT2_KILL;
      else if (gun_available(c, mun, lf, step)) {
        mun -= 1; lf = step;
        io.st(TE_D_MUNITION, s, (uint32_t)mun); io.st(TE_D_LAST_FIRED, s, (uint32_t)step);
        const U4 r = env_rng(c, env, RNG_HIT, (uint32_t)s, 0, episode, (uint32_t)step);
        if (u01(r.x) < c.hit_prob) rec |= SLOT2_KILL;
      }
    }
    if (a_me && tgt >= 0 && dmin < c.explosion_range) rec |= SLOT_EXPLODE;
    *reinterpret_cast<uint2*>(&sm[R.prec2(s) * 64 + 2 * lane]) = make_uint2(rec, __float_as_uint(dmin));
  }
  TE_SLOT_BARRIER();

  // ---- P3
  const uint32_t S = L(R.accS()), zone = L(R.accZone());
  uint32_t killed = 0u;
  int shots = 0, exploded = 0;
  float cur = 0.0f;   // stage02_agent_min_distance: the first armed pursuer's closest invader (0 when there is none), before any respawn
  bool found = false;
  for (int q = 0; q < P; ++q) {
    const uint2 r = *reinterpret_cast<const uint2*>(&sm[R.prec2(q) * 64 + 2 * lane]);
    const int t = (int)(r.x & 0xFFu) - 1;
    if (!found && ((S >> q) & 1u)) { found = true; cur = t >= 0 ? __uint_as_float(r.y) : 0.0f; }
    if (r.x & SLOT2_KILL) { killed |= 1u << t; shots += 1; }
    if (r.x & SLOT_EXPLODE) { killed |= (1u << q) | (1u << t); exploded += 1; }
  }
  uint32_t A = S & ~killed;
  if ((killed >> s) & 1u) io.disarm(s);
  const bool outside_p = (zone & pur_bits) != 0u, outside_i = (zone & inv_bits) != 0u;
  const bool term = step > max_step || outside_p || outside_i || __popc(A & pur_bits) < P;
  const bool to_terminal = valid && term && c.auto_reset;
  bool own = false;
  if (s >= 1) {   // the observation is taken before the respawn: a drone armed after the step broadcast has no Delta = 1 snapshot yet
    own = ((A >> s) & 1u) != 0u && rh < 1.0f;
    for (uint32_t m = all_bits & ~1u & ~(1u << s); m; m &= m - 1u) {
      const int k = __ffs((int)m) - 1;
      const uint2 cr = *reinterpret_cast<const uint2*>(&sm[R.cellr(k) * 64 + 2 * lane]);
      const float rk = __uint_as_float(cr.y);
      const bool beaten = ((A >> k) & 1u) != 0u && cr.x == (uint32_t)cj && (rk < rh || (rk == rh && k < s));
      own = own && !beaten;
    }
    if (own && valid) Lor(R.accOwn(), 1u << s);
  }
  const float flag_me = (float)(s < P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) / 5.0f;
  const bool time_plane = c.lidar_channels != 2;
  auto patch = [&](float* dst) {
    dst += (size_t)env * lidar_words(c);
    dst[cj] = rh; dst[TE_LIDAR_CELLS + cj] = flag_me;
    if (time_plane) dst[2 * TE_LIDAR_CELLS + cj] = 0.1f;
  };
  if (valid && own && !to_terminal && o.obs.lidar) patch(o.obs.lidar);
  const unsigned long long term_b = __ballot(to_terminal);
  if (o.term.lidar && term_b) {
    for (unsigned long long tb = term_b; tb; tb &= tb - 1) {
      const int l = __ffsll((long long)tb) - 1;
      float* tile = o.term.lidar + (size_t)(blockIdx.x * 64 + l) * lidar_words(c);
      for (int e = s * 64 + lane; e < lidar_words(c); e += 64 * D) tile[e] = 1.0f;
    }
  }
  if (s == 0) {   // compute_reward (stages.py:241-300), outputs, env record
    if (valid) {
      io.stef(TE_E_LAST_ACTION + 0, act.x); io.stef(TE_E_LAST_ACTION + 1, act.y); io.stef(TE_E_LAST_ACTION + 2, act.z); io.stef(TE_E_LAST_ACTION + 3, act.w);
      io.ste(TE_E_STEP, (uint32_t)step); io.ste(TE_E_SNAP_MASK, S);
    }
    kills += shots; deads += exploded;
    float gs[3];
    gun_state(c, mun, lf, step, max_munition_of(c, 0), gs);
    const bool ready = gs[2] == 1.0f || gs[0] == 0.0f;
    float bonus = 0.0f, penalty = 0.0f;
    const float score = ready ? -cur : cur * (2.0f * gs[1] - 1.0f);
    if (0.01f < last - cur && ready) bonus += c.approach_bonus_gain * fnorm(V3{ag[3], ag[4], ag[5]});
    bonus += 1000.0f * (float)shots; penalty += 1000.0f * (float)exploded;
    if (outside_p) penalty += 1000.0f;
    const float reward = score + bonus - penalty;
    if (valid) {
      io.ste(TE_E_AGENT_KILLS, (uint32_t)kills); io.ste(TE_E_DEADS, (uint32_t)deads);
      o.reward[env] = reward; o.done[env] = term ? 1 : 0;
      reinterpret_cast<int4*>(o.info)[env] = make_int4(kills, 0, deads, 0);
    }
    if (to_terminal) {
      if (o.term.inertial) inertial_row_regs(c, o.term.inertial + (size_t)env * TE_OBS_INERTIAL_WORDS, mx, my, mz, ag, mun, lf, step);
      if (o.term.last_action) reinterpret_cast<float4*>(o.term.last_action)[env] = act;
    }
    if (valid) { io.stef(TE_E_PREV_SNAP_MIN, cur); io.stef(TE_E_LAST_DIST, cur); }   // on_step_end: last_offsets = current_offsets
  }
  // on_step_end: killed invaders come back (stages.py:167-174); an auto-reset respawns every drone (L3Stage1.on_reset, stages.py:104-131)
  if (valid && !is_p && !((A >> s) & 1u)) {
    const V3 w = stage02_invader_position(c, env, s, episode, (uint32_t)step);
    io.respawn(c, s, w); mx = w.x; my = w.y; mz = w.z;
  }
  if (to_terminal) {
    episode += 1u; step = 0;
    if (s == 0) {
      io.ste(TE_E_EPISODE, episode); io.ste(TE_E_STEP, 0u); io.ste(TE_E_MAX_STEP, (uint32_t)c.max_step); io.ste(TE_E_ROUND, 0u);
      io.ste(TE_E_AGENT_KILLS, 0u); io.ste(TE_E_ALLIES_KILLS, 0u); io.ste(TE_E_DEADS, 0u);
#pragma unroll
      for (int k = 0; k < 4; ++k) io.ste(TE_E_LAST_ACTION + k, 0u);
      io.ste(TE_E_SNAP_MASK, all_bits);
    }
    V3 w;
    if (is_p) {
      const U4 r = env_rng(c, env, RNG_SPAWN_PURSUER, (uint32_t)s, 0, episode, 0);
      w = stage02_position(c.pursuer_spawn_radius, 0.0f, u01(r.x), u01(r.y), u01(r.z));
      mun = max_munition_of(c, s); lf = -c.cooldown_steps;
    } else w = stage02_invader_position(c, env, s, episode, 0u);
    io.respawn(c, s, w); mx = w.x; my = w.y; mz = w.z;
    act = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) ag[k] = 0.0f;
  }
  L(R.npos(0, s)) = __float_as_uint(mx); L(R.npos(1, s)) = __float_as_uint(my); L(R.npos(2, s)) = __float_as_uint(mz);
  if (o.term.lidar && term_b) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TE_SLOT_BARRIER();

  // ---- P4
  if (s >= 1 && to_terminal && own && o.term.lidar) patch(o.term.lidar);
  if (o.persist && o.obs.lidar && valid) {
    const uint32_t owners = L(R.accOwn());
    uint16_t* pv = o.prev + env;
    if (s == 0) pv[0] = (uint16_t)(to_terminal ? 0 : __popc(owners));
    else if (own && !to_terminal) pv[(size_t)(__popc(owners & ((1u << s) - 1u)) + 1) * p.Npad] = (uint16_t)cj;
  }
  if (s == 0) {
    if (to_terminal) {   // closest invader of pursuer 0 on the fresh positions
      float d0 = 0.0f; bool any = false;
      for (int j = P; j < D; ++j) {
        const float d = fdist(V3{mx, my, mz}, V3{Lf(R.npos(0, j)), Lf(R.npos(1, j)), Lf(R.npos(2, j))});
        const bool take = !any || d < d0;
        d0 = take ? d : d0; any = true;
      }
      io.stef(TE_E_PREV_SNAP_MIN, d0); io.stef(TE_E_LAST_DIST, d0);
    }
    if (o.obs.inertial && valid) inertial_row_regs(c, o.obs.inertial + (size_t)env * TE_OBS_INERTIAL_WORDS, mx, my, mz, ag, mun, lf, step);
    if (valid && o.obs.last_action) reinterpret_cast<float4*>(o.obs.last_action)[env] = act;
  }
  if (s == D - 1) {   // every armed slot flies as a dense wave outside the level4 family; the invaders are all armed again
    const uint32_t armed_post = to_terminal ? all_bits : ((A & pur_bits) | (valid ? inv_bits : 0u));
    uint64_t dense = 0u;
    for (int k = 0; k < D; ++k) if (__ballot(valid && ((armed_post >> k) & 1u))) dense |= (uint64_t)1 << k;
    if (lane == 0) { p.slot_mask[blockIdx.x] = dense; p.mixed_count[blockIdx.x] = 0u; p.live_mask[blockIdx.x] = dense; }
  }
}

}  // namespace te
