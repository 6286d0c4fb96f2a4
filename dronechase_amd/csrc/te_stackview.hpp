// te_stackview.hpp — level5's stacked observation in two launches of single-purpose waves (round 2), replacing the phase chain of
// stacked_kernel (te_stacked.hpp: one 512-thread workgroup per 64 envs, ~20 barriers, every operand through LDS, 2 workgroups per CU):
//
//   ring_push_kernel<DM>   one wave per (64-env chunk, wingman), lane = env.  The wingman's own sphere — every other armed drone re-framed,
//                          binned, closer wins per cell — lives in registers (compile-time slot capacity DM, run-time predicates) and goes
//                          straight into this step's ring entry.  No LDS, no barrier.                      (stacked_kernel phase 1)
//   stack_view_kernel<DM>  one 5-wave workgroup per chunk for one observer: wave 0 draws the neighbourhood and lists the observer's own
//                          entry, waves 1-4 re-project one drawn snapshot each (farther wins) into a list of their own in LDS; then stack
//                          order + validity mask, ones into the terminal tiles of auto-reset envs, the patches, and the ring reset of
//                          auto-reset envs.  4 barriers.                                                   (phases 2, 3, 4)
//
// Same arithmetic, same helpers (spherical_of, rotate_by, draw_stack, ring layout) and the same order of every farther- / closer-wins
// decision as stacked_kernel, which stays as the fallback for shapes beyond DM = 37 and behind TE_STACKED=lds.
#pragma once
#include <type_traits>
#include <utility>

#include "te_stacked.hpp"

namespace te {

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop the front end expands
template <class F, int... Is>
TE_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
TE_DEV void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// W > 1 (small shards: fewer (chunk, wingman) pairs than the chip has SIMDs, every push wave alone on its SIMD): the binning — 200 of the
// wave's 250 instructions per drone — is dealt over the W waves of a workgroup (slot j to wave j mod W), handed over in LDS rows, and wave 0
// alone runs closer-wins and writes the entry: a 12 us chain becomes ~5.  Same functions on the same operands: the entry is bit for bit the
// one-wave kernel's.  Large shards keep W = 1 (the launch is VALU-bound there, the hand-over would only add to it).
template <int DM, int W>
__global__ __launch_bounds__(64 * W) void ring_push_kernel(StackParams p) {
  extern __shared__ uint32_t push_sm[];   // W > 1: 4 * DM rows of 64 words
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers;
  const SnapRows sr{D, P};
  const int chunk = blockIdx.x / P, pp = blockIdx.x - chunk * P;
  const int lane = threadIdx.x & 63, env = chunk * kEPB + lane;
  const int w = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  // env < Npad always (the planes are padded to whole chunks), so every lane may read; the lanes beyond N contribute nothing
  auto snap = [&](int w) { return p.snap[(size_t)w * p.Npad + env]; };
  const bool live = env < p.N;
  const uint64_t A = live ? ((uint64_t)snap(sr.armed()) | ((uint64_t)snap(sr.armed_hi()) << 32)) : 0ull;
  const bool mine = live && ((A >> pp) & 1u);
  // Slots at or above `hi` are disarmed in every publishing env of the wave (a round arms the first invaders of the table: 5 to 15 of
  // level5_fusion's 30 in its first rounds): the compile-time loops below skip them on a scalar test.  The butterfly runs BEFORE any lane
  // leaves: __shfl_xor is ds_bpermute under EXEC, and an inactive source lane would hand back 0 instead of relaying its partial maximum.
  int top = mine ? 64 - __clzll((unsigned long long)A) : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) top = max(top, __shfl_xor(top, off));
  const int hi = __builtin_amdgcn_readfirstlane(top);
  if (!mine) return;   // a disarmed wingman publishes nothing: its ring keeps the older entries
  const int step = (int)snap(sr.step());
  const V3 me{__uint_as_float(snap(sr.pos() + 0 * D + pp)), __uint_as_float(snap(sr.pos() + 1 * D + pp)), __uint_as_float(snap(sr.pos() + 2 * D + pp))};
  const Q4 q = quat_of_euler(V3{__uint_as_float(snap(sr.euler() + 0 * P + pp)), __uint_as_float(snap(sr.euler() + 1 * P + pp)), __uint_as_float(snap(sr.euler() + 2 * P + pp))});
  const Q4 qi = inverse_of(q);
  // ---- the sphere of wingman pp: (r_hat, theta, phi, cell) of every other armed drone.  The slot loops are expanded by the front end
  // (static_for: one lambda instance per compile-time index), not by the loop unroller: at DM = 37 the 37 x 37 closer-wins loop is
  // beyond the unroller's budget, it stayed a loop, the per-slot arrays were indexed at run time and went to scratch (640 B, round 2).
  float rh[DM], th[DM], ph[DM]; uint32_t cell[DM];
  static_for<DM>([&](auto J) {
    constexpr int j = decltype(J)::value;
    cell[j] = 0xFFFFFFFFu; rh[j] = 1.0f; th[j] = 0.0f; ph[j] = 0.0f;
    if (W > 1 && (j % W) != w) return;   // another wave's slot
    if (j < hi && j != pp && ((A >> j) & 1u)) {
      const V3 pj{__uint_as_float(snap(sr.pos() + 0 * D + j)), __uint_as_float(snap(sr.pos() + 1 * D + j)), __uint_as_float(snap(sr.pos() + 2 * D + j))};
      const V3 local = rotate_by(qi, sub(pj, me));
      int cl;
      spherical_of(c, local, rh[j], th[j], ph[j], cl);
      cell[j] = (uint32_t)cl;
    }
  });
  if (W > 1) {   // hand-over: every wave its slots' four words, then wave 0 collects
    static_for<DM>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if ((j % W) != w || j >= hi) return;
      push_sm[(0 * DM + j) * 64 + lane] = __float_as_uint(rh[j]); push_sm[(1 * DM + j) * 64 + lane] = __float_as_uint(th[j]);
      push_sm[(2 * DM + j) * 64 + lane] = __float_as_uint(ph[j]); push_sm[(3 * DM + j) * 64 + lane] = cell[j];
    });
    __syncthreads();
    if (w != 0) return;
    static_for<DM>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if ((j % W) == 0 || j >= hi) return;
      rh[j] = __uint_as_float(push_sm[(0 * DM + j) * 64 + lane]); th[j] = __uint_as_float(push_sm[(1 * DM + j) * 64 + lane]);
      ph[j] = __uint_as_float(push_sm[(2 * DM + j) * 64 + lane]); cell[j] = push_sm[(3 * DM + j) * 64 + lane];
    });
  }
  // ---- closer wins per cell (lidar_math.py:262-311), as in stacked_kernel: j CLAIMS its cell if it is in view with r_hat < 1; the kept
  // feature of a cell is its closest claimant (ties: lowest slot); cells appear in the entry in the order of their first claimant.
  // Two drones in one of the sphere's 338 cells are rare (a pair collides in 4 % of the waves), and every pair was paying for the full
  // bookkeeping (7 VALU instructions x DM^2 pairs: 2 600 of the wave's 5 700 at DM = 18).  Now a pair costs ONE compare and a scalar branch:
  // a drone that claims nothing gets a code no other slot has, so "same cell" is equality of the codes, and only a pair that does collide in
  // some lane enters its bookkeeping: the later one is not the cell's first claimant (`later`), the farther one (ties: the higher slot) is
  // `beaten`.  A cell's kept feature is its one unbeaten claimant: the first claimant itself, or — found by a search only a wave with
  // such a lane runs — another one.
  uint32_t* ent = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)env, pp, step);
  using M = typename std::conditional<(DM > 32), uint64_t, uint32_t>::type;   // one bit per slot
  constexpr M one = 1;
  M later = 0, beaten = 0;
  static_for<DM>([&](auto J) {
    constexpr int j = decltype(J)::value;
    cell[j] = (cell[j] != 0xFFFFFFFFu && rh[j] < 1.0f) ? cell[j] : (0xFFFF0000u | (uint32_t)j);   // from here on: the code
  });
  static_for<DM>([&](auto J) {
    constexpr int j = decltype(J)::value;
    if (j >= hi) return;   // wave-uniform
    static_for<j>([&](auto K) {
      constexpr int k = decltype(K)::value;
      const bool same = cell[k] == cell[j];
      if (__ballot(same) != 0ull) {
        const bool k_closer = rh[k] <= rh[j];   // ties: the lower slot k stays
        later |= same ? (one << j) : (M)0;
        beaten |= same ? (k_closer ? (one << j) : (one << k)) : (M)0;
      }
    });
  });
  uint32_t n = 0u;
  static_for<DM>([&](auto J) {
    constexpr int j = decltype(J)::value;
    if (j >= hi) return;   // wave-uniform
    const bool lead = cell[j] < 0xFFFF0000u && !((later >> j) & one);
    float br = rh[j], bt = th[j], bp = ph[j]; int bw = j;
    if (__ballot(lead && ((beaten >> j) & one)) != 0ull) {   // the first claimant is not the closest one: the cell's unbeaten claimant
      static_for<DM>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if (k <= j) return;   // (an earlier claimant of the cell would have made j a later one)
        if (lead && ((beaten >> j) & one) && cell[k] == cell[j] && !((beaten >> k) & one)) { br = rh[k]; bt = th[k]; bp = ph[k]; bw = k; }
      });
    }
    if (lead) {
      const uint32_t meta = (uint32_t)(bw < P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) | ((uint32_t)bw << 8);
      *reinterpret_cast<uint4*>(ent + TE_RING_HEADER_WORDS + 4 * n) = make_uint4(__float_as_uint(br), __float_as_uint(bt), __float_as_uint(bp), meta);
      n += 1u;
    }
  });
  *reinterpret_cast<uint4*>(ent) = make_uint4((uint32_t)step, n, __float_as_uint(me.x), __float_as_uint(me.y));
  *reinterpret_cast<uint4*>(ent + 4) = make_uint4(__float_as_uint(me.z), __float_as_uint(q.x), __float_as_uint(q.y), __float_as_uint(q.z));
  *reinterpret_cast<uint4*>(ent + 8) = make_uint4(__float_as_uint(q.w), 0u, 0u, 0u);
}

// LDS of stack_view_kernel: five lists (own, four neighbours) of up to F = D - 1 (cell | type << 16, r_hat) pairs per env, then the rows below
struct ViewRows {
  int F;
  TE_DEV int list(int k) const { return k * 2 * F; }        // k = 0 own, 1..4 neighbours
  TE_DEV int n(int k) const { return 10 * F + k; }          // kept features of list k, or 0xFFFFFFFF = no sphere
  TE_DEV int dn() const { return 10 * F + 5; }
  TE_DEV int dwho() const { return dn() + 1; }
  TE_DEV int dage() const { return dn() + 2; }
  TE_DEV int dperm() const { return dn() + 3; }
  TE_DEV int step() const { return dn() + 4; }
  TE_DEV int done() const { return dn() + 5; }
  TE_DEV int opos() const { return dn() + 6; }              // 5
  TE_DEV int total() const { return opos() + 5; }
};
__host__ __device__ inline int view_lds_rows(int D) { return 10 * (D - 1) + 5 + 6 + 5; }
constexpr int kViewThreads = 5 * 64;
constexpr int kPushSplit = 4;   // waves that share the binning of one (chunk, wingman) pair in ring_push_kernel<DM, W> on small shards
constexpr int kViewBatch = 8;   // ring features a wave of stack_view_kernel requests at once

template <int DM>
__global__ __launch_bounds__(kViewThreads) void stack_view_kernel(StackParams p, StackOut o) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers, ob = p.observer;
  const SnapRows sr{D, P};
  const ViewRows r{D - 1};
  const int env0 = blockIdx.x * kEPB, nvalid = min(kEPB, p.N - env0);
  const int tid = threadIdx.x, l = tid & (kEPB - 1), role = tid >> 6;   // role 0: draws + own list; 1..4: neighbour role - 1
  const int env = env0 + l;
  const bool valid = l < nvalid;
  auto row = [&](int rr, int ll) -> uint32_t& { return sm[rr * kEPB + ll]; };
  auto rowf = [&](int rr, int ll) { return __uint_as_float(sm[rr * kEPB + ll]); };
  auto snap = [&](int w) { return p.snap[(size_t)w * p.Npad + env]; };
  auto orow = [&](int ll) { return (size_t)(env0 + ll) * (size_t)p.n_obs + (size_t)ob; };
  // ---- (2) the observer's own snapshot and its draws (wave 0), with the own sphere listed out of the ring
  if (role == 0) {
    uint32_t n = 0u, who = 0u, age = 0u, perm = 0u, own_n = 0xFFFFFFFFu;
    const int step = valid ? (int)snap(sr.step()) : 0;
    const uint32_t armed = valid ? snap(sr.armed()) : 0u;
    bool own_ok = valid && step >= 1;
    if (own_ok) {
      const uint32_t* own = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)env, ob, step);
      if (p.push) own_ok = ((armed >> ob) & 1u) != 0u;            // pushed by ring_push_kernel iff the observer is armed
      else own_ok = (int)load_fresh(own) == step;                 // te_observe_stacked: whatever the ring holds
      if (own_ok) {
        const int cnt = min((int)load_fresh(own + 1), r.F);   // (a ring that came in through te_set_state cannot overrun the list)
        for (int k0 = 0; k0 < cnt; k0 += kViewBatch) {
          te_w4 fb[kViewBatch];
#pragma unroll
          for (int i = 0; i < kViewBatch; ++i) fb[i] = load_fresh4(own + TE_RING_HEADER_WORDS + 4 * min(k0 + i, r.F - 1));
#pragma unroll
          for (int i = 0; i < kViewBatch; ++i) {
          const int k = k0 + i;
          if (k >= cnt) continue;
          const te_w4 f = fb[i];
          const float thv = __uint_as_float(f.y), phv = __uint_as_float(f.z);
          const int ti = min(max((int)(thv / kPi * (float)TE_LIDAR_NTHETA), 0), TE_LIDAR_NTHETA - 1);
          const int pi = min(max((int)((phv + kPi) / (2.0f * kPi) * (float)TE_LIDAR_NPHI), 0), TE_LIDAR_NPHI - 1);
          row(r.list(0) + 2 * k, l) = (uint32_t)(ti * TE_LIDAR_NPHI + pi) | ((f.w & 0xFFu) << 16);
          row(r.list(0) + 2 * k + 1, l) = f.x;
          }
        }
        own_n = (uint32_t)cnt;
      }
    }
    if (own_ok) {
      int nn;
      draw_stack(c, env, ob, valid ? snap(sr.episode()) : 0u, (uint32_t)step, armed & ((1u << P) - 1u), P, nn, who, age, perm);
      n = (uint32_t)nn;
    }
    row(r.n(0), l) = own_n;   // 0xFFFFFFFF: _build_valid_spheres returns [] without an own snapshot (fused_lidar.py:91-96)
    row(r.dn(), l) = n; row(r.dwho(), l) = who; row(r.dage(), l) = age; row(r.dperm(), l) = perm;
    row(r.step(), l) = (uint32_t)step; row(r.done(), l) = valid ? snap(sr.done()) : 0u;
  }
  __syncthreads();
  // ---- (3) neighbour k of env l: transform_features + add_features(invert) (lidar_math.py:186-345), sequentially, one wave per k
  if (role >= 1) {
    const int k = role - 1;
    uint32_t count = 0xFFFFFFFFu;
    if (valid && k < (int)row(r.dn(), l)) {
      const int step = (int)row(r.step(), l);
      const int q = (int)((row(r.dwho(), l) >> (8 * k)) & 0xFFu), a = (int)((row(r.dage(), l) >> (8 * k)) & 0xFFu);
      const int s = step - (a - 1);
      const uint32_t* nb = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)env, q, s > 0 ? s : 0);
      if (s >= 1 && (int)load_fresh(nb) == s) {  // get_snapshot -> None otherwise (lidar_buffer.py:152-154)
        const uint32_t* own = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)env, ob, step);
        const te_w4 n0 = load_fresh4(nb), n1 = load_fresh4(nb + 4), n2 = load_fresh4(nb + 8), o0 = load_fresh4(own), o1 = load_fresh4(own + 4), o2 = load_fresh4(own + 8);
        const V3 pn{__uint_as_float(n0.z), __uint_as_float(n0.w), __uint_as_float(n1.x)};
        const Q4 qn{__uint_as_float(n1.y), __uint_as_float(n1.z), __uint_as_float(n1.w), __uint_as_float(n2.x)};
        const V3 po{__uint_as_float(o0.z), __uint_as_float(o0.w), __uint_as_float(o1.x)};
        const Q4 qo{__uint_as_float(o1.y), __uint_as_float(o1.z), __uint_as_float(o1.w), __uint_as_float(o2.x)};
        const M3 Rn = rotation(qn), Ro = rotation(inverse_of(qo));
        const int cnt = min((int)n0.y, r.F);
        const int base = r.list(1 + k);
        count = 0u;
        // (the features are requested kViewBatch at a time, unconditionally — index clamped into the entry — before the first one is looked at:
        // one request per iteration made a neighbour's list a chain of up to D - 1 memory round trips, most of this wave's time on a small shard)
        for (int f0 = 0; f0 < cnt; f0 += kViewBatch) {
          te_w4 fb[kViewBatch];
#pragma unroll
          for (int i = 0; i < kViewBatch; ++i) fb[i] = load_fresh4(nb + TE_RING_HEADER_WORDS + 4 * min(f0 + i, r.F - 1));
#pragma unroll
          for (int i = 0; i < kViewBatch; ++i) {
          if (f0 + i >= cnt) continue;
          const te_w4 ft = fb[i];
          const uint32_t meta = ft.w;
          if ((int)((meta >> 8) & 0xFFu) == ob) continue;  // synthetic echo of the observer itself (lidar_math.py:228-232)
          const float R = __uint_as_float(ft.x) * c.lidar_radius, thv = __uint_as_float(ft.y), phv = __uint_as_float(ft.z);
          const float st = sin_rev(thv * (0.5f / kPi)), ct = cos_rev(thv * (0.5f / kPi)), sp = sin_rev(phv * (0.5f / kPi)), cp = cos_rev(phv * (0.5f / kPi));
          const V3 cart{R * st * cp, R * st * sp, R * ct};                          // spherical_to_cartesian (lidar_math.py:16-22)
          const V3 glob = mul(Rn, cart);
          const V3 loc = mul(Ro, V3{glob.x + pn.x - po.x, glob.y + pn.y - po.y, glob.z + pn.z - po.z});
          float rhat, t2, p2; int cl;
          spherical_of(c, loc, rhat, t2, p2, cl);
          int at = -1;
          for (uint32_t j = 0; j < count; ++j) if ((row(base + 2 * j, l) & 0xFFFFu) == (uint32_t)cl) at = (int)j;
          bool put = true;
          if (at >= 0) { const float cur = rowf(base + 2 * at + 1, l); put = cur < 1.0f ? rhat > cur : true; }
          else at = (int)count++;
          if (put) { row(base + 2 * at, l) = (uint32_t)cl | ((meta & 0xFFu) << 16); row(base + 2 * at + 1, l) = __float_as_uint(rhat); }
          }
        }
      }
    }
    row(r.n(1 + k), l) = count;
  }
  __syncthreads();
  // ---- (4a) stack order -> output positions and the validity mask (fused_lidar.py:293-326,246-262)
  if (role == 0 && valid) {
    const bool done = row(r.done(), l) != 0u;
    uint8_t* M = (done ? o.t_mask : o.mask);
    const bool any = row(r.n(0), l) != 0xFFFFFFFFu;
    int sidx[5]; int nv = 0;   // stack index of own = 0, of valid neighbour k = 1 + (valid neighbours before k)
    sidx[0] = any ? nv++ : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) sidx[1 + k] = (any && row(r.n(1 + k), l) != 0xFFFFFFFFu) ? nv++ : -1;
    const uint32_t perm = row(r.dperm(), l);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      uint32_t at = 0xFFu;
      if (sidx[s] >= 0) for (int i = 0; i < TE_STACK_SPHERES; ++i) if ((int)((perm >> (4 * i)) & 0xFu) == sidx[s]) at = (uint32_t)i;
      row(r.opos() + s, l) = at;
    }
    if (M) for (int i = 0; i < TE_STACK_SPHERES; ++i) M[orow(l) * TE_STACK_SPHERES + i] = (any && (int)((perm >> (4 * i)) & 0xFu) < nv) ? 1 : 0;
    if (done && o.mask) for (int i = 0; i < TE_STACK_SPHERES; ++i) o.mask[orow(l) * TE_STACK_SPHERES + i] = 0;  // reset observation
  }
  // ---- (4b) terminal tiles of auto-reset envs are not pre-filled: ones first (rare, block-uniform test)
  const bool lane_done = role == 0 && valid && row(r.done(), l) != 0u;
  if (__syncthreads_or(lane_done ? 1 : 0)) {
    if (o.t_stacked)
      for (int ll = 0; ll < nvalid; ++ll) {
        if (!row(r.done(), ll)) continue;
        float* base = o.t_stacked + orow(ll) * TE_OBS_STACKED_WORDS;
        for (int e = tid; e < TE_OBS_STACKED_WORDS; e += kViewThreads) base[e] = 1.0f;
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- (4c) patches: every wave its own list (own sphere: time = 1/10, perception_snapshot.py:36-37; neighbour: normalized_delta of the snapshot)
  if (valid) {
    const uint32_t n = row(r.n(role), l);
    const uint32_t at = row(r.opos() + role, l);
    const bool done = row(r.done(), l) != 0u;
    float* basep = done ? o.t_stacked : o.stacked;
    const bool mine = n != 0xFFFFFFFFu && at != 0xFFu;
    if (p.persist && o.stacked) {
      // Persistent observation: the main buffer is all ones when this launch starts — streamed by the sub-step launch's fill waves
      // (persist == 2), or, when it still held the previous observation, its recorded cells set back to one by that launch's erase waves
      // (persist == 1).  Here the cells patched now are RECORDED per output sphere for the next call; the spheres nobody writes (padding;
      // all six of an auto-reset env, whose reset observation is empty) get an empty record from the wave whose role is the sphere's rank
      // among the unused ones (mod 5).  With several observers per env (te_step_students, persist == 3) the erase happens here instead, by
      // the wave that owns the sphere: 43 008 erase waves in front of the flights cost that path more than they hid (3.47 vs 2.75 ms per step).
      uint16_t* pv = p.prev + (size_t)ob * TE_STACK_SPHERES * (size_t)D * p.Npad + env;
      uint32_t used = 0u;
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) { const uint32_t a5 = row(r.opos() + s5, l); if (!done && row(r.n(s5), l) != 0xFFFFFFFFu && a5 != 0xFFu) used |= 1u << a5; }
      int rank = 0;
      for (int sphere = 0; sphere < TE_STACK_SPHERES; ++sphere) {
        const bool is_mine = !done && mine && (int)at == sphere;
        const bool unused = !((used >> sphere) & 1u);
        const bool take = is_mine || (unused && (rank % 5) == role);
        if (unused) rank += 1;
        if (!take) continue;
        uint16_t* ps = pv + (size_t)sphere * D * p.Npad;
        if (p.persist == 3) {   // te_step_students: the owner of a sphere erases it here (same wave as the patches below: stores stay in issue order)
          const int cnt = (int)ps[0];
          float* d0 = o.stacked + orow(l) * TE_OBS_STACKED_WORDS + (size_t)sphere * TE_OBS_LIDAR_WORDS;
          for (int i = 0; i < cnt; ++i) { float* d = d0 + ps[(size_t)(1 + i) * p.Npad]; d[0] = 1.0f; d[TE_LIDAR_CELLS] = 1.0f; d[2 * TE_LIDAR_CELLS] = 1.0f; }   // (cell by cell: plane by plane re-reads the list, 2.99 vs 2.70 ms per te_step_students)
        }
        if (!is_mine) ps[0] = 0;
      }
    }
    if (mine && basep) {
      float* d0 = basep + orow(l) * TE_OBS_STACKED_WORDS + (size_t)at * TE_OBS_LIDAR_WORDS;
      const float tnorm = role == 0 ? 0.1f : (float)((row(r.dage(), l) >> (8 * (role - 1))) & 0xFFu) / (float)TE_RING_DEPTH;
      const int base = r.list(role);
      // plane by plane (see engage_kernel: consecutive stores that stay within one plane of the tile drain faster)
      for (uint32_t f = 0; f < n; ++f) d0[row(base + 2 * f, l) & 0xFFFFu] = rowf(base + 2 * f + 1, l);
      for (uint32_t f = 0; f < n; ++f) { const uint32_t w = row(base + 2 * f, l); d0[TE_LIDAR_CELLS + (w & 0xFFFFu)] = (float)(w >> 16) / 5.0f; }
      for (uint32_t f = 0; f < n; ++f) d0[2 * TE_LIDAR_CELLS + (row(base + 2 * f, l) & 0xFFFFu)] = tnorm;
      if (p.persist && o.stacked && !done) {
        uint16_t* ps = p.prev + ((size_t)ob * TE_STACK_SPHERES + at) * (size_t)D * p.Npad + env;
        ps[0] = (uint16_t)n;
        for (uint32_t f = 0; f < n; ++f) ps[(size_t)(1 + f) * p.Npad] = (uint16_t)(row(base + 2 * f, l) & 0xFFFFu);
      }
    }
  }
  // ---- (4d) buffer reset of auto-reset envs (base_lidar.py:62-66: step 0 resets every LIDAR buffer): after every wave has read the ring
  if (p.push) {
    __syncthreads();
    for (int it = tid; it < kEPB * P * TE_RING_DEPTH; it += kViewThreads) {
      const int ll = it & (kEPB - 1), e = it / kEPB;
      if (ll >= nvalid || !row(r.done(), ll)) continue;
      p.ring[(((size_t)(env0 + ll) * P) * TE_RING_DEPTH + (size_t)e) * (size_t)p.entry_words] = 0u;
    }
  }
}

}  // namespace te
