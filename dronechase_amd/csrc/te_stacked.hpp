// te_stacked.hpp — level5: the per-wingman snapshot ring and the FusedLIDAR stacked observation of the agent
// (reference: core/entities/quadcopters/components/sensors/fused_lidar.py:73-109,143-262,293-326;
//  components/lidar_buffer.py:10-157; components/lidar_math.py:186-345; threatsense/level5/level5_envrionment.py:312-351).
//
// Canonical clock (SEMANTICS.md): the ring entry of env-step s holds a wingman's IMU pose of step s and the kept
// features of its own sphere built from the poses of step s; at step t the entry of step t has age 1
// (normalized_delta 0.1, as in the own sphere) and an entry of age a is the one of step t - a + 1.
//
// The engage/observe kernel leaves a *snapshot* of what the observation of this step may look at (IMU positions of
// all drones, IMU attitude of the wingmen, who is armed after the engagement, step / episode / done) in a few
// planes BEFORE it respawns anything; stacked_kernel (one 512-thread block per 64 envs) then
//   (1) pushes every armed wingman's entry into the ring          one (env, other drone) item per thread, per wingman
//   (2) draws the agent's neighbourhood (Philox)                  one lane per env
//   (3) re-projects each chosen snapshot into the agent's frame   one (env, neighbour) per thread, farther wins
//   (4) patches own + neighbour spheres into the [N,6,3,13,26] buffer that the sub-step kernel's fill waves
//       already set to ones, writes the validity mask, and clears the ring of auto-reset envs.
#pragma once
#include "te_logic.hpp"

namespace te {

// snapshot planes written by the engage/observe kernel: word w of env e at snap[w * Npad + e]
struct SnapRows {
  int D, P;
  TE_DEV int pos() const { return 0; }                 // 3*D : IMU positions, word-major, slot-minor
  TE_DEV int euler() const { return 3 * D; }           // 3*P : IMU attitude of the wingmen
  TE_DEV int armed() const { return 3 * D + 3 * P; }   // bit s = drone s armed after the engagement
  TE_DEV int step() const { return armed() + 1; }      // RL step of the observation (before any auto-reset)
  TE_DEV int episode() const { return armed() + 2; }
  TE_DEV int done() const { return armed() + 3; }      // the env auto-resets in this step
  TE_DEV int armed_hi() const { return armed() + 4; }  // bits 32.. of the armed mask (more than 32 drones per env)
  TE_DEV int total() const { return armed() + 5; }
};
__host__ __device__ inline int snap_words(int D, int P) { return 3 * D + 3 * P + 5; }

struct StackOut { float* stacked; uint8_t* mask; float* t_stacked; uint8_t* t_mask; };

// LDS rows (kEPB words each) of stacked_kernel
struct StackRows {
  int D, P;
  TE_DEV int F() const { return D - 1; }                      // most features one entry can keep
  TE_DEV int pos() const { return 0; }                        // 3*D
  TE_DEV int quat() const { return 3 * D; }                   // 4*P : quaternion rebuilt from the IMU attitude (imu.py:38)
  TE_DEV int armed() const { return quat() + 4 * P; }         // 1
  TE_DEV int step() const { return armed() + 1; }
  TE_DEV int episode() const { return armed() + 2; }
  TE_DEV int done() const { return armed() + 3; }
  TE_DEV int armed_hi() const { return armed() + 4; }         // bits 32.. of the armed mask (more than 32 drones per env)
  // phase (1) uses feat + work, phase (3) re-uses the same rows for the four neighbour lists (157 >= 136 rows at D = 18):
  // the block stays below 80 KB of LDS, i.e. two blocks per CU
  TE_DEV int feat() const { return armed() + 5; }             // 4*D : r_hat, theta, phi, cell of drone j seen from wingman p
  TE_DEV int work() const { return feat() + 4 * D; }          // 5*F : closer-wins working list (cell, r_hat, theta, phi, meta)
  TE_DEV int nb_k() const { return feat(); }                  // 4 * 2*F (aliases feat + work): cell | type << 16, r_hat
  TE_DEV int own_n() const { return work() + 5 * F(); }       // 1   : kept features of the agent's own sphere
  TE_DEV int own_k() const { return own_n() + 1; }            // 2*F : cell | type << 16, r_hat
  TE_DEV int dn() const { return own_k() + 2 * F(); }         // 1   : neighbours drawn
  TE_DEV int dwho() const { return dn() + 1; }                // 1   : 4 x 8 bits
  TE_DEV int dage() const { return dn() + 2; }                // 1   : 4 x 8 bits
  TE_DEV int dperm() const { return dn() + 3; }               // 1   : 6 x 4 bits, out[i] = stack[perm[i]]
  TE_DEV int nb_n() const { return dn() + 4; }                // 4   : kept features of neighbour k, or 0xFFFFFFFF = no sphere
  TE_DEV int opos() const { return nb_n() + 4; }              // 5   : output position of own / neighbour k (or 0xFF)
  TE_DEV int total() const { return opos() + 5; }
};
__host__ __device__ inline int stack_lds_rows(int D, int P) { return 3 * D + 4 * P + 5 + 4 * D + 5 * (D - 1) + 1 + 2 * (D - 1) + 4 + 4 + 5; }
static_assert(4 * 32 + 5 * 31 >= 8 * 31 && 4 * 64 + 5 * 63 >= 8 * 63, "the neighbour lists must fit in feat + work");

TE_DEV V3 rotate_by(Q4 q, V3 v) { return mul(rotation(q), v); }  // pybullet rotateVector
TE_DEV Q4 inverse_of(Q4 q) {  // LidarMath._invert_quaternion (lidar_math.py:40-51)
  const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  return Q4{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
}
// cartesian -> (r_hat, theta, phi, cell)  (lidar_math.py:24-34,93-96,128-137), on the native rsq / rcp and the polynomial asin / atan2 of the
// sub-step loop (1e-7 abs; te_engage.hpp: lidar_cell_fast): libm's acosf + atan2f were most of the instructions of phases (1) and (3)
TE_DEV void spherical_of(const te_config& c, V3 v, float& rhat, float& theta, float& phi, int& cell) {
  const float r2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float r = 0.0f;
  theta = 0.0f; phi = 0.0f;
  if (r2 != 0.0f) {
    const float inv = rsq(r2);
    r = r2 * inv;
    theta = 0.5f * kPi - fast_asin(clampf(v.z * inv, -1.0f, 1.0f));
    phi = fast_atan2(v.y, v.x);
  }
  rhat = clampf(r * rcp(c.lidar_radius), 0.0f, 1.0f);
  const int ti = min(max((int)(theta / kPi * (float)TE_LIDAR_NTHETA), 0), TE_LIDAR_NTHETA - 1);
  const int pi = min(max((int)((phi + kPi) / (2.0f * kPi) * (float)TE_LIDAR_NPHI), 0), TE_LIDAR_NPHI - 1);
  cell = ti * TE_LIDAR_NPHI + pi;
}
TE_DEV uint32_t* ring_entry_ptr(uint32_t* ring, int entry_words, int P, size_t env, int p, int step) {
  return ring + ((env * (size_t)P + (size_t)p) * TE_RING_DEPTH + (size_t)(step % TE_RING_DEPTH)) * (size_t)entry_words;
}
// the ring is written and read by different waves of one block: read around the (write-through, not coherent) L1
TE_DEV uint32_t load_fresh(const uint32_t* p) { return __builtin_nontemporal_load(p); }
typedef uint32_t te_w4 __attribute__((ext_vector_type(4)));
TE_DEV te_w4 load_fresh4(const uint32_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const te_w4*>(p)); }   // entries and features are 16-byte aligned

// (2) FusedLIDAR.bootstrap / randomize_stack draws (fused_lidar.py:73-81,246-262; lidar_buffer.py:110-157): n ~ U{1..4};
// n distinct armed wingmen (the agent included) without replacement; an age ~ U{1..9} each; a uniform permutation of
// the six (sphere, valid) pairs.  16 Philox words keyed (STACK, slot 0, sub 0..3, episode, step); an integer in
// [0, k) is (word * k) >> 32.  The test checker restates exactly this sequence.
TE_DEV void draw_stack(const te_config& c, int env, int observer, uint32_t episode, uint32_t step, uint32_t armed_pursuers, int P, int& n,
                       uint32_t& who, uint32_t& age, uint32_t& perm) {
  uint32_t w[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const U4 r = env_rng(c, env, RNG_STACK, (uint32_t)observer, (uint32_t)k, episode, step);   // every observer draws its own neighbourhood
    w[4 * k] = r.x; w[4 * k + 1] = r.y; w[4 * k + 2] = r.z; w[4 * k + 3] = r.w;
  }
  const uint32_t cand = armed_pursuers;  // candidates as a bit set; "position i of the array" = i-th set bit still unused
  const int nc = __popc(cand);
  const int want = 1 + (int)(((uint64_t)w[0] * 4u) >> 32);
  n = want < nc ? want : nc;
  // Partial Fisher-Yates over the candidates in slot order.  The array lives in registers: position x holds the x-th set bit of `cand`
  // unless one of the (at most four) swaps moved something there — a dynamically indexed local array is scratch memory, and this
  // phase runs on one wave of the block while the other seven wait (14 -> 3 us of every block).
  uint32_t ov_pos = 0xFFFFFFFFu, ov_val = 0u;           // up to 4 overrides of positions >= the draw index: (position, value) bytes
  auto nth = [&](int x) {                                // slot of the x-th set bit of cand
    uint32_t m = cand;
    for (int k = 0; k < x; ++k) m &= m - 1u;
    return (uint32_t)(__ffs((int)m) - 1);
  };
  auto get = [&](int x) {
    uint32_t v = nth(x);
#pragma unroll
    for (int k = 0; k < 4; ++k) if ((int)((ov_pos >> (8 * k)) & 0xFFu) == x) v = (ov_val >> (8 * k)) & 0xFFu;   // later overrides win: scanned in order
    return v;
  };
  who = 0u; age = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < n) {
      const int j = i + (int)(((uint64_t)w[1 + i] * (uint32_t)(nc - i)) >> 32);
      const uint32_t ai = get(i), aj = get(j);
      // arr[i] <-> arr[j]: position i is never read again (the next draws start at i + 1); position j now holds the old arr[i]
      ov_pos = (ov_pos & ~(0xFFu << (8 * i))) | ((uint32_t)j << (8 * i));
      ov_val = (ov_val & ~(0xFFu << (8 * i))) | (ai << (8 * i));
      who |= aj << (8 * i);
      age |= (uint32_t)(1 + (int)(((uint64_t)w[5 + i] * (uint32_t)(TE_RING_DEPTH - 1)) >> 32)) << (8 * i);
    }
  }
  // uniform permutation of the six (sphere, valid) pairs: Fisher-Yates on 4-bit fields of one register
  perm = 0x543210u;
#pragma unroll
  for (int i = TE_STACK_SPHERES - 1; i >= 1; --i) {
    const int j = (int)(((uint64_t)w[9 + (TE_STACK_SPHERES - 1 - i)] * (uint32_t)(i + 1)) >> 32);
    const uint32_t vi = (perm >> (4 * i)) & 0xFu, vj = (perm >> (4 * j)) & 0xFu;
    perm = (perm & ~(0xFu << (4 * i))) | (vj << (4 * i));
    perm = (perm & ~(0xFu << (4 * j))) | (vi << (4 * j));
  }
}

struct StackParams {
  te_config cfg;
  const uint32_t* snap;  // snapshot planes
  uint32_t* ring;
  int N, Npad, D, entry_words;
  int push;              // 1: te_step_stacked (push this step's entries, clear the ring of auto-reset envs); 0: te_observe_stacked
  int observer;          // whose FusedLIDAR.read_data this launch serves: 0 = the agent; te_step_students launches it once per wingman
  int n_obs;             // observers per env in the output buffers: 1 ([N,6,...]) or P ([N,P,6,...], te_step_students)
  // persistent observation (te_set_persistent_obs; stack_view_kernel only).  0: patch the buffer the fill waves of this step's sub-step
  // launch streamed ones over.  2: the same, and RECORD which cells of every output sphere were patched.  1: the buffer still held the
  // previous call's observation and the sub-step launch's erase waves (FillJob mode 3) have set its recorded cells back to one: patch and
  // record — nothing else of the 24 KB per env is touched.  3 (several observers per env, te_step_students): like 1, but the wave that
  // owns an output sphere erases it in stack_view_kernel itself.
  // prev[((observer * 6 + sphere) * D + i) * Npad + env]: i = 0 the count, i = 1.. the cells (u16, env fastest: lanes read coalesced)
  int persist;
  uint16_t* prev;
};

#ifdef TE_DEBUG_STAMPS   // phase stamps of every workgroup (tools/stacked_stamps.py), in the records the engage kernel's stamps used before this launch
#define TE_SSTAMP(idx) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (g_te_dbg && threadIdx.x == 0) g_te_dbg[64 + blockIdx.x * 16 + (idx)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TE_SSTAMP(idx) do {} while (0)
#endif
constexpr int kStackThreads = 512;  // 8 waves: the block's LDS (73 KB at D = 18) allows two blocks per CU, i.e. 4 waves per SIMD (+ 9 % env-steps/s over 256 threads)
__global__ __launch_bounds__(kStackThreads) void stacked_kernel(StackParams p, StackOut o) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  const te_config& c = p.cfg;
  const int D = p.D, P = c.n_pursuers;
  const StackRows r{D, P};
  const SnapRows sr{D, P};
  const int env0 = blockIdx.x * kEPB, nvalid = min(kEPB, p.N - env0);
  const int tid = threadIdx.x, lane = tid & (kEPB - 1);
  const int ob = p.observer;
  TE_SSTAMP(0);
  auto orow = [&](int l) { return (size_t)(env0 + l) * (size_t)p.n_obs + (size_t)ob; };   // row of (env, observer) in the output buffers
  auto row = [&](int rr, int l) -> uint32_t& { return sm[rr * kEPB + l]; };
  auto rowf = [&](int rr, int l) { return __uint_as_float(sm[rr * kEPB + l]); };
  // ---- stage the snapshot (coalesced: planes are env-fastest)
  for (int it = tid; it < kEPB * sr.total(); it += blockDim.x) {
    const int l = it & (kEPB - 1), w = it / kEPB;
    const uint32_t v = p.snap[(size_t)w * p.Npad + env0 + l];
    if (w < 3 * D) row(r.pos() + w, l) = v;
    else if (w < sr.armed()) row(r.feat() + (w - 3 * D), l) = v;  // euler parked in the feature rows until the quaternions are built
    else row(r.armed() + (w - sr.armed()), l) = v;
  }
  __syncthreads();
  for (int it = tid; it < kEPB * P; it += blockDim.x) {  // quaternion of every wingman, rounded to float32 as the snapshot does
    const int l = it & (kEPB - 1), pp = it / kEPB;
    const Q4 q = quat_of_euler(V3{rowf(r.feat() + 0 * P + pp, l), rowf(r.feat() + 1 * P + pp, l), rowf(r.feat() + 2 * P + pp, l)});
    row(r.quat() + 0 * P + pp, l) = __float_as_uint(q.x); row(r.quat() + 1 * P + pp, l) = __float_as_uint(q.y);
    row(r.quat() + 2 * P + pp, l) = __float_as_uint(q.z); row(r.quat() + 3 * P + pp, l) = __float_as_uint(q.w);
  }
  if (tid < kEPB) { row(r.own_n(), tid) = 0u; }
  __syncthreads();
  TE_SSTAMP(1);
  auto pos_of = [&](int s, int l) { return V3{rowf(r.pos() + 0 * D + s, l), rowf(r.pos() + 1 * D + s, l), rowf(r.pos() + 2 * D + s, l)}; };
  auto quat_of = [&](int pp, int l) { return Q4{rowf(r.quat() + 0 * P + pp, l), rowf(r.quat() + 1 * P + pp, l), rowf(r.quat() + 2 * P + pp, l), rowf(r.quat() + 3 * P + pp, l)}; };

  // ---- (1) FusedLIDAR.update_data of every armed wingman -> ring entry of this step
  if (p.push) {
    for (int pp = 0; pp < P; ++pp) {
      for (int it = tid; it < kEPB * D; it += blockDim.x) {
        const int l = it & (kEPB - 1), j = it / kEPB;
        const uint64_t A = (uint64_t)row(r.armed(), l) | ((uint64_t)row(r.armed_hi(), l) << 32);
        if (j == pp || !((A >> j) & 1u) || !((A >> pp) & 1u)) { row(r.feat() + 3 * D + j, l) = 0xFFFFFFFFu; continue; }
        const V3 local = rotate_by(inverse_of(quat_of(pp, l)), sub(pos_of(j, l), pos_of(pp, l)));
        float rhat, th, ph; int cell;
        spherical_of(c, local, rhat, th, ph, cell);
        row(r.feat() + 0 * D + j, l) = __float_as_uint(rhat); row(r.feat() + 1 * D + j, l) = __float_as_uint(th);
        row(r.feat() + 2 * D + j, l) = __float_as_uint(ph); row(r.feat() + 3 * D + j, l) = (uint32_t)cell;
      }
      __syncthreads();
      // closer wins per cell (lidar_math.py:262-311), evaluated per (env, drone j) item instead of walking a list:
      //   j is a CLAIMANT of its cell if it is in view with r_hat < 1 (an empty cell holds 1.0, strict '<');
      //   the kept feature of a cell is its closest claimant (ties: lowest slot, as a later equal range does not replace);
      //   the cells appear in the entry in the order of their first claimant (the reference's dict insertion order).
      // Step (i): leader flag (first claimant of its cell) and the cell's winner; step (ii): position = leaders before me.
      const int lead = r.work(), win = r.work() + D;
      for (int it = tid; it < kEPB * D; it += blockDim.x) {
        const int l = it & (kEPB - 1), j = it / kEPB;
        const uint32_t cell = row(r.feat() + 3 * D + j, l);
        uint32_t is_lead = 0u, w = (uint32_t)j;
        if (cell != 0xFFFFFFFFu && rowf(r.feat() + j, l) < 1.0f) {
          is_lead = 1u;
          float best = rowf(r.feat() + j, l);
          for (int k = 0; k < D; ++k) {
            if (k == j || row(r.feat() + 3 * D + k, l) != cell) continue;
            const float rk = rowf(r.feat() + k, l);
            if (!(rk < 1.0f)) continue;
            if (k < j) is_lead = 0u;
            if (rk < best || (rk == best && k < (int)w)) { best = rk; w = (uint32_t)k; }
          }
        }
        row(lead + j, l) = is_lead; row(win + j, l) = w;
      }
      __syncthreads();
      for (int it = tid; it < kEPB * D; it += blockDim.x) {
        const int l = it & (kEPB - 1), j = it / kEPB;
        if (l >= nvalid || !((row(r.armed(), l) >> pp) & 1u)) continue;
        const int step = (int)row(r.step(), l);
        uint32_t* ent = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)(env0 + l), pp, step);
        if (j == pp) {  // this item has no feature of its own: it writes the header
          uint32_t n = 0u;
          for (int k = 0; k < D; ++k) n += row(lead + k, l);
          const V3 me = pos_of(pp, l); const Q4 q = quat_of(pp, l);
          *reinterpret_cast<uint4*>(ent) = make_uint4((uint32_t)step, n, __float_as_uint(me.x), __float_as_uint(me.y));
          *reinterpret_cast<uint4*>(ent + 4) = make_uint4(__float_as_uint(me.z), __float_as_uint(q.x), __float_as_uint(q.y), __float_as_uint(q.z));
          *reinterpret_cast<uint4*>(ent + 8) = make_uint4(__float_as_uint(q.w), 0u, 0u, 0u);
          if (pp == ob) row(r.own_n(), l) = n;
        } else if (row(lead + j, l)) {
          uint32_t at = 0u;
          for (int k = 0; k < j; ++k) at += row(lead + k, l);
          const int w = (int)row(win + j, l);
          const uint32_t meta = (uint32_t)(w < P ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION) | ((uint32_t)w << 8);
          const uint4 f = make_uint4(row(r.feat() + 0 * D + w, l), row(r.feat() + 1 * D + w, l), row(r.feat() + 2 * D + w, l), meta);
          *reinterpret_cast<uint4*>(ent + TE_RING_HEADER_WORDS + 4 * at) = f;   // slots >= n keep stale words: never read
          if (pp == ob) {  // the observer keeps (cell, type, r_hat) of its own sphere for the patch phase
            row(r.own_k() + 2 * at, l) = row(r.feat() + 3 * D + j, l) | ((meta & 0xFFu) << 16); row(r.own_k() + 2 * at + 1, l) = f.x;
          }
        }
      }
      __syncthreads();
    }
  }
  TE_SSTAMP(2);
  // ---- (2) the agent's own snapshot and its draws
  if (tid < kEPB) {
    const int l = tid;
    uint32_t n = 0u, who = 0u, age = 0u, perm = 0u;
    const int step = (int)row(r.step(), l);
    bool own_ok = l < nvalid && step >= 1;
    if (own_ok) {
      const uint32_t* own = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)(env0 + l), ob, step);
      if (p.push) own_ok = ((row(r.armed(), l) >> ob) & 1u) != 0u;  // pushed above iff the observer is armed
      else {  // te_observe_stacked: the own sphere comes back out of the ring
        own_ok = (int)load_fresh(own) == step;
        if (own_ok) {
          const int cnt = (int)load_fresh(own + 1);
          for (int k = 0; k < cnt; ++k) {
            const uint32_t* f = own + TE_RING_HEADER_WORDS + 4 * k;
            float rhat = __uint_as_float(load_fresh(f)), th = __uint_as_float(load_fresh(f + 1)), ph = __uint_as_float(load_fresh(f + 2));
            const int ti = min(max((int)(th / kPi * (float)TE_LIDAR_NTHETA), 0), TE_LIDAR_NTHETA - 1);
            const int pi = min(max((int)((ph + kPi) / (2.0f * kPi) * (float)TE_LIDAR_NPHI), 0), TE_LIDAR_NPHI - 1);
            row(r.own_k() + 2 * k, l) = (uint32_t)(ti * TE_LIDAR_NPHI + pi) | ((load_fresh(f + 3) & 0xFFu) << 16);
            row(r.own_k() + 2 * k + 1, l) = __float_as_uint(rhat);
          }
          row(r.own_n(), l) = (uint32_t)cnt;
        }
      }
    }
    if (own_ok) {
      int nn;
      draw_stack(c, env0 + l, ob, row(r.episode(), l), (uint32_t)step, row(r.armed(), l) & ((1u << P) - 1u), P, nn, who, age, perm);
      n = (uint32_t)nn;
    } else row(r.own_n(), l) = 0xFFFFFFFFu;  // _build_valid_spheres returns [] without an own snapshot (fused_lidar.py:91-96)
    row(r.dn(), l) = n; row(r.dwho(), l) = who; row(r.dage(), l) = age; row(r.dperm(), l) = perm;
  }
  __threadfence_block();
  __syncthreads();
  TE_SSTAMP(3);
  // ---- (3) neighbour k of env l: transform_features + add_features(invert) (lidar_math.py:186-345), sequentially
  {
    const int l = lane, k = tid >> 6;
    uint32_t count = 0xFFFFFFFFu;
    if (l < nvalid && k < (int)row(r.dn(), l)) {
      const int step = (int)row(r.step(), l);
      const int q = (int)((row(r.dwho(), l) >> (8 * k)) & 0xFFu), a = (int)((row(r.dage(), l) >> (8 * k)) & 0xFFu);
      const int s = step - (a - 1);
      const uint32_t* nb = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)(env0 + l), q, s > 0 ? s : 0);
      if (s >= 1 && (int)load_fresh(nb) == s) {  // get_snapshot -> None otherwise (lidar_buffer.py:152-154)
        const uint32_t* own = ring_entry_ptr(p.ring, p.entry_words, P, (size_t)(env0 + l), ob, step);
        // one 16-byte load per feature and three per header instead of a dword load per word (each a round trip past the L1)
        const te_w4 n0 = load_fresh4(nb), n1 = load_fresh4(nb + 4), n2 = load_fresh4(nb + 8), o0 = load_fresh4(own), o1 = load_fresh4(own + 4), o2 = load_fresh4(own + 8);
        const V3 pn{__uint_as_float(n0.z), __uint_as_float(n0.w), __uint_as_float(n1.x)};
        const Q4 qn{__uint_as_float(n1.y), __uint_as_float(n1.z), __uint_as_float(n1.w), __uint_as_float(n2.x)};
        const V3 po{__uint_as_float(o0.z), __uint_as_float(o0.w), __uint_as_float(o1.x)};
        const Q4 qo{__uint_as_float(o1.y), __uint_as_float(o1.z), __uint_as_float(o1.w), __uint_as_float(o2.x)};
        const M3 Rn = rotation(qn), Ro = rotation(inverse_of(qo));
        const int cnt = (int)n0.y;
        const int base = r.nb_k() + k * 2 * r.F();
        count = 0u;
        for (int f = 0; f < cnt; ++f) {
          const te_w4 ft = load_fresh4(nb + TE_RING_HEADER_WORDS + 4 * f);
          const uint32_t meta = ft.w;
          if ((int)((meta >> 8) & 0xFFu) == ob) continue;  // synthetic echo of the observer itself (lidar_math.py:228-232)
          const float R = __uint_as_float(ft.x) * c.lidar_radius, th = __uint_as_float(ft.y), ph = __uint_as_float(ft.z);
          const float st = sin_rev(th * (0.5f / kPi)), ct = cos_rev(th * (0.5f / kPi)), sp = sin_rev(ph * (0.5f / kPi)), cp = cos_rev(ph * (0.5f / kPi));   // v_sin / v_cos take revolutions
          const V3 cart{R * st * cp, R * st * sp, R * ct};                          // spherical_to_cartesian (lidar_math.py:16-22)
          const V3 glob = mul(Rn, cart);
          const V3 loc = mul(Ro, V3{glob.x + pn.x - po.x, glob.y + pn.y - po.y, glob.z + pn.z - po.z});
          float rhat, t2, p2; int cell;
          spherical_of(c, loc, rhat, t2, p2, cell);
          int at = -1;
          for (uint32_t j = 0; j < count; ++j) if ((row(base + 2 * j, l) & 0xFFFFu) == (uint32_t)cell) at = (int)j;
          bool put = true;
          if (at >= 0) { const float cur = rowf(base + 2 * at + 1, l); put = cur < 1.0f ? rhat > cur : true; }
          else at = (int)count++;
          if (put) { row(base + 2 * at, l) = (uint32_t)cell | ((meta & 0xFFu) << 16); row(base + 2 * at + 1, l) = __float_as_uint(rhat); }
        }
      }
    }
    if (k < 4) row(r.nb_n() + k, l) = count;
  }
  __syncthreads();
  TE_SSTAMP(4);
  // ---- (4a) stack order -> output positions and the validity mask (fused_lidar.py:293-326,246-262)
  if (tid < kEPB && tid < nvalid) {
    const int l = tid;
    const bool done = row(r.done(), l) != 0u;
    uint8_t* M = (done ? o.t_mask : o.mask);
    const bool any = row(r.own_n(), l) != 0xFFFFFFFFu;
    // stack index of own = 0, of valid neighbour k = 1 + (valid neighbours before k)
    int sidx[5]; int nv = 0;
    sidx[0] = any ? nv++ : -1;
    for (int k = 0; k < 4; ++k) sidx[1 + k] = (any && row(r.nb_n() + k, l) != 0xFFFFFFFFu) ? nv++ : -1;
    const uint32_t perm = row(r.dperm(), l);
    for (int s = 0; s < 5; ++s) {
      uint32_t at = 0xFFu;
      if (sidx[s] >= 0) for (int i = 0; i < TE_STACK_SPHERES; ++i) if ((int)((perm >> (4 * i)) & 0xFu) == sidx[s]) at = (uint32_t)i;
      row(r.opos() + s, l) = at;
    }
    if (M) for (int i = 0; i < TE_STACK_SPHERES; ++i) M[orow(l) * TE_STACK_SPHERES + i] = (any && (int)((perm >> (4 * i)) & 0xFu) < nv) ? 1 : 0;
    if (done && o.mask) for (int i = 0; i < TE_STACK_SPHERES; ++i) o.mask[orow(l) * TE_STACK_SPHERES + i] = 0;  // reset observation
  }
  // ---- (4b) terminal tiles of auto-reset envs are not pre-filled: ones first (rare, block-uniform test)
  const bool lane_done = tid < kEPB && tid < nvalid && row(r.done(), tid) != 0u;
  if (__syncthreads_or(lane_done ? 1 : 0)) {
    if (o.t_stacked)
      for (int l = 0; l < nvalid; ++l) {
        if (!row(r.done(), l)) continue;
        float* base = o.t_stacked + orow(l) * TE_OBS_STACKED_WORDS;
        for (int e = tid; e < TE_OBS_STACKED_WORDS; e += blockDim.x) base[e] = 1.0f;
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  TE_SSTAMP(5);
  // ---- (4c) patches
  auto dst_of = [&](int l, int sphere_slot) -> float* {
    const uint32_t at = row(r.opos() + sphere_slot, l);
    float* basep = row(r.done(), l) ? o.t_stacked : o.stacked;
    if (at == 0xFFu || !basep) return nullptr;
    return basep + orow(l) * TE_OBS_STACKED_WORDS + (size_t)at * TE_OBS_LIDAR_WORDS;
  };
  for (int it = tid; it < kEPB * (D - 1); it += blockDim.x) {  // own sphere: time = 1/10 (perception_snapshot.py:36-37)
    const int l = it & (kEPB - 1), f = it / kEPB;
    if (l >= nvalid) continue;
    const uint32_t n = row(r.own_n(), l);
    if (n == 0xFFFFFFFFu || f >= (int)n) continue;
    float* d = dst_of(l, 0);
    if (!d) continue;
    const uint32_t w = row(r.own_k() + 2 * f, l);
    d += (w & 0xFFFFu);
    d[0] = rowf(r.own_k() + 2 * f + 1, l); d[TE_LIDAR_CELLS] = (float)(w >> 16) / 5.0f; d[2 * TE_LIDAR_CELLS] = 0.1f;
  }
  {
    const int l = lane, k = tid >> 6;
    const uint32_t n = (l < nvalid && k < 4) ? row(r.nb_n() + k, l) : 0xFFFFFFFFu;
    if (n != 0xFFFFFFFFu) {
      float* d0 = dst_of(l, 1 + k);
      if (d0) {
        const float tnorm = (float)((row(r.dage(), l) >> (8 * k)) & 0xFFu) / (float)TE_RING_DEPTH;  // normalized_delta of the snapshot
        const int base = r.nb_k() + k * 2 * r.F();
        for (uint32_t f = 0; f < n; ++f) {
          const uint32_t w = row(base + 2 * f, l);
          float* d = d0 + (w & 0xFFFFu);
          d[0] = rowf(base + 2 * f + 1, l); d[TE_LIDAR_CELLS] = (float)(w >> 16) / 5.0f; d[2 * TE_LIDAR_CELLS] = tnorm;
        }
      }
    }
  }
  TE_SSTAMP(6);
  // ---- (4d) buffer reset of auto-reset envs (base_lidar.py:62-66: step 0 resets every LIDAR buffer)
  if (p.push)
    for (int it = tid; it < kEPB * P * TE_RING_DEPTH; it += blockDim.x) {
      const int l = it & (kEPB - 1), e = it / kEPB;
      if (l >= nvalid || !row(r.done(), l)) continue;
      p.ring[(((size_t)(env0 + l) * P) * TE_RING_DEPTH + (size_t)e) * (size_t)p.entry_words] = 0u;
    }
  TE_SSTAMP(7);
}

}  // namespace te
