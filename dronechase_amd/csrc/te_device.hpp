// te_device.hpp — device-side building blocks of the gfx950 threat-engagement environment.
//
// Data layout in HBM (struct-of-arrays "planes", env index fastest):
//   drone word w of slot s of env e :  dstate[(w * D + s) * Npad + e]      (w < TE_DRONE_WORDS + TE_X_WORDS)
//   env   word w of env e           :  estate[w * Npad + e]               (w < TE_ENV_WORDS)
// Npad = N rounded up to 64, so one wavefront (64 lanes = 64 consecutive envs of ONE slot) reads and
// writes every plane with perfectly coalesced 256-byte accesses, and "the other drones of my env" are
// at the same lane offset of another plane (no shuffles, no LDS needed for neighbour data).
//
// Reference citations are file:line under the reference's src/ tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/threatengage.h"

namespace te {

// extra per-drone planes that are not part of the public state blob
enum { TE_X_CMD = TE_DRONE_WORDS,     /* 3: velocity command vx,vy,vz of a scripted ALLY for the next step */
       TE_X_REF = TE_DRONE_WORDS + 3, /* 3: a PURSUER's IMU position as the last engage/observe launch (or reset) left it: what the
                                         invaders' navigators steer at.  Not TE_D_OBS_POS itself: the pursuers' waves overwrite that
                                         at the end of the very sub-step launch in which the invaders' waves read it */
       TE_X_WORDS = 6 };

// constants of the sub-step loop derived from te_config.  Computed ONCE on the host (te_create) and passed
// by value in the kernel arguments, so they live in SGPRs: gfx950 has no scalar float ALU, and deriving
// them in the kernel would pin ~20 VGPRs of wave-uniform values for the whole loop.
struct Derived {
  float invT, dt, k_motor, noise_ratio, q_thrust, q_arm, q_torque, k_drag, k_pqr;
  float dt_inv_ix, dt_inv_iy, dt_inv_iz, ix, iy, iz, dt_inv_m, dt_g, pwm_floor;
  // PID gains with the controller period folded in: ki * T and kd / T
  float lv_kiT[2], lv_kdT[2], av_kiT[3], av_kdT[3], zv_kiT, zv_kdT;
  float half_dt, quarter_dt2, noise_m2ln2;
  int ctrl_ratio;  // physics sub-steps per controller period (control_dt / physics_dt = 2): only read when cfg.control_every_substep == 0
  float ground_rest;  // z of a hull resting on the ground plane (ground_z + hull_half_height); used when cfg.ground_contact  // dt/2, dt^2/4, -2 ln 2 * noise_ratio^2 (Box-Muller radius incl. the noise gain)
};
constexpr int kMixedWaves = 2, kMixedCap = 64 * kMixedWaves;  // mixed waves per chunk; a slot that would overflow the list flies densely
constexpr int kDenseMin = 40;                                // default of Params::dense_min: armed envs of a chunk from which a slot gets its own wave
// (member order matters: the launch arguments are this 1 KB struct by value, and a wave's first requests need the POINTERS — with the two
// configuration blocks in front of them the sub-step launch of an 8 192-env shard took 18.5 instead of 17.9 us, the engage launch 9.5 instead of 9.3)
struct Params {
  uint32_t* dstate;
  uint32_t* estate;
  int N, Npad, D;
  // staged row r of the engage/observe kernel lives at word stage_tab[r] & 0x7FFFFFFF of dstate (bit 31: of estate),
  // env-fastest: filled once by te_create, read with scalar loads (the row index is wave-uniform)
  uint32_t* stage_tab;
  // bit s of slot_mask[chunk] = some env of that chunk of 64 has drone s armed: written by the engage/observe launch
  // (and by reset / set_state), read by the sub-step launch with a SCALAR load, so that a wave with nothing to fly
  // retires without a single vector memory operation
  uint64_t* slot_mask;  // [Npad / 64] (bit s for slot s < 64)
  uint64_t* live_mask;  // [Npad / 64] bit s = SOME env of the chunk has slot s armed (dense or mixed): the rows the next engage launch requests
  // Slots armed in only a few envs of a chunk do not get a wave of their own (at 4.2 armed drones per env a rollout has
  // 7.6 armed slots per chunk, i.e. 56 % of the lanes of the flight waves would idle): their (env, slot) items share
  // MIXED waves of 64 items.  slot_mask then holds the slots flown densely; mixed_items[chunk * kMixedCap + i] =
  // lane | slot << 8 for i < mixed_count[chunk], written together with slot_mask.
  int dense_min;           // kDenseMin; TE_DENSE_MIN=1 turns the mixed waves off (every armed slot flies densely)
  uint32_t* mixed_count;   // [Npad / 64]
  uint16_t* mixed_items;   // [Npad / 64][kMixedCap]
  // level5 (cfg.stacked_obs): observation-time snapshot planes (te_stacked.hpp SnapRows) and the snapshot ring; else null
  uint32_t* snap;
  uint32_t* ring;
  int entry_words;
  unsigned long long* dbg;  // phase stamps of one block (diagnostic builds with -DTE_DEBUG_STAMPS only; else unused)
  te_config cfg;
  Derived kd;
};
#ifdef TE_DEBUG_STAMPS
__device__ unsigned long long* g_te_dbg = nullptr;
#endif
// (the LDS engage/observe kernel's own stamps need -DTE_LDS_STAMPS on top: next to this round's StepOut the compiler (ROCm 7.2) rejects the
// flat null test of (p).dbg in those kernels with "Illegal instruction detected: V_CMP_NE_U32 $src_shared_base"; tools/k2_blocks.py, k2_stamps.py)
#if defined(TE_DEBUG_STAMPS) && defined(TE_LDS_STAMPS) && !defined(TE_NO_STAMP)
#define TE_STAMP(p, blk, idx)                                                                     \
  do {                                                                                            \
    if ((p).dbg && threadIdx.x == 0) {                                                            \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                             \
      if (blockIdx.x == (blk)) (p).dbg[(idx)] = t_;                                               \
      (p).dbg[64 + blockIdx.x * 16 + (idx)] = t_; /* every block: tools/k2_blocks.py */            \
    }                                                                                             \
  } while (0)
#else
#define TE_STAMP(p, blk, idx) do {} while (0)
#endif
// finer stamps from inside the per-env logic (lane 0 of wave 0 of every block), slots 8..15 of the block's record
#if defined(TE_DEBUG_STAMPS) && !defined(TE_NO_LSTAMP)
#define TE_LSTAMP(idx)                                                                            \
  do {                                                                                            \
    if (g_te_dbg && threadIdx.x == 0) g_te_dbg[64 + blockIdx.x * 16 + (idx)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define TE_LSTAMP(idx) do {} while (0)
#endif

#define TE_DEV __device__ __forceinline__

constexpr float kPi = 3.14159265358979323846f;

// ---------------------------------------------------------------- plane accessors
struct Planes {
  uint32_t* d; uint32_t* e; int D; int Npad; int env;
  TE_DEV float& df(int w, int s) const { return reinterpret_cast<float*>(d)[((size_t)w * D + s) * Npad + env]; }
  TE_DEV int32_t& di(int w, int s) const { return reinterpret_cast<int32_t*>(d)[((size_t)w * D + s) * Npad + env]; }
  TE_DEV float& ef(int w) const { return reinterpret_cast<float*>(e)[(size_t)w * Npad + env]; }
  TE_DEV int32_t& ei(int w) const { return reinterpret_cast<int32_t*>(e)[(size_t)w * Npad + env]; }
};

// One drone slot of one env for the sub-step kernel, through buffer instructions: a 128-bit resource per state
// array (SGPRs), ONE shared 32-bit VGPR byte offset per lane, the plane as the scalar offset.  Written with the
// raw-buffer builtins because the equivalent pointer arithmetic only sometimes lowers to "saddr + voffset":
// when it does not, every plane gets its own 64-bit VGPR address pair and the kernel loses two waves per SIMD.
// te_create rejects state arrays beyond 2^32 bytes; out-of-range offsets read 0 / drop the store.
struct SlotLane {
  __amdgpu_buffer_rsrc_t d, e;
  uint32_t plane_bytes, off_bytes, eplane_bytes, env_bytes;
  TE_DEV SlotLane(uint32_t* dstate, uint32_t* estate, uint32_t D, uint32_t npad, uint32_t slot, uint32_t env, uint32_t d_words,
                  uint32_t e_words)
      : d(__builtin_amdgcn_make_buffer_rsrc(dstate, 0, (int)(d_words * 4u), 0x00020000)),
        e(__builtin_amdgcn_make_buffer_rsrc(estate, 0, (int)(e_words * 4u), 0x00020000)),
        plane_bytes(D * npad * 4u), off_bytes((slot * npad + env) * 4u), eplane_bytes(npad * 4u), env_bytes(env * 4u) {}
  TE_DEV float lf(int w) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d, (int)off_bytes, (int)((uint32_t)w * plane_bytes), 0)); }
  TE_DEV int32_t li(int w) const { return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(d, (int)off_bytes, (int)((uint32_t)w * plane_bytes), 0); }
  TE_DEV void sf(int w, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), d, (int)off_bytes, (int)((uint32_t)w * plane_bytes), 0); }
  TE_DEV void si(int w, int32_t v) const { __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v, d, (int)off_bytes, (int)((uint32_t)w * plane_bytes), 0); }
  // word w of ANOTHER slot of this lane's env (cross-drone reads of the scripted navigators)
  TE_DEV float lf_slot(int w, int s) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d, (int)env_bytes, (int)((uint32_t)w * plane_bytes + (uint32_t)s * eplane_bytes), 0));
  }
  TE_DEV int32_t lei(int w) const { return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(e, (int)env_bytes, (int)((uint32_t)w * eplane_bytes), 0); }
};

// ---------------------------------------------------------------- Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
// 32x32 -> 64-bit product in ONE quarter-rate instruction (hipcc emits v_mul_hi_u32 + v_mul_lo_u32, two of them)
TE_DEV void mul_hi_lo(uint32_t a, uint32_t m, uint32_t& hi, uint32_t& lo) {
  uint64_t prod;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(prod) : "v"(a), "s"(m) : "vcc");
  hi = (uint32_t)(prod >> 32); lo = (uint32_t)prod;
}
// a ^ b ^ k in ONE instruction: gfx950's three-input bit operation with the truth table of a three-way xor (the compiler emits two v_xor)
TE_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t k) {
#ifdef TE_NO_XOR3   // A/B builds only (tools/ab.sh)
  return a ^ b ^ k;
#else
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
#endif
}
// Philox4x32-R (Salmon et al., SC'11; Random123).  R = 10 everywhere a draw decides something (spawns, hits, neighbourhoods, actions);
// R = 7 — the smallest round count the paper reports as Crush-resistant, with its own known-answer vectors in Random123 — for the motor
// noise, which is 16 of a drone's 16 sub-step draws per env-step and ~12 % of the sub-step loop's issue cycles at R = 10.  The keys are
// wave-uniform (cfg.seed).
template <int ROUNDS>
TE_DEV U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    mul_hi_lo(c0, 0xD2511F53u, hi0, lo0);
    mul_hi_lo(c2, 0xCD9E8D57u, hi1, lo1);
    uint32_t n0 = xor3(hi1, c1, k0), n2 = xor3(hi0, c3, k1);
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
TE_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) { return philox4x32<10>(c0, c1, c2, c3, k0, k1); }
#ifndef TE_MOTOR_ROUNDS
#define TE_MOTOR_ROUNDS 7   // (-DTE_MOTOR_ROUNDS=10: A/B builds only; the oracle draws with 7)
#endif
constexpr int kMotorNoiseRounds = TE_MOTOR_ROUNDS;
enum { RNG_SPAWN_INVADER = 1, RNG_SPAWN_PURSUER = 2, RNG_HIT = 3, RNG_MOTOR = 4, RNG_ACTION = 5, RNG_RESPAWN = 6, RNG_STACK = 7 };
// counter = { global env (low 32), purpose | slot<<8 | sub<<16 | global env (high 8)<<24, episode, index }
TE_DEV U4 env_rng(const te_config& c, int env, uint32_t purpose, uint32_t slot, uint32_t sub, uint32_t episode,
                  uint32_t index) {
  uint64_t g = (uint64_t)c.env_index_base + (uint64_t)env;
  return philox4x32_10((uint32_t)g, purpose | (slot << 8) | (sub << 16) | ((uint32_t)(g >> 32) << 24), episode, index,
                       (uint32_t)c.seed, (uint32_t)(c.seed >> 32));
}
TE_DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
TE_DEV float u01_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// ---------------------------------------------------------------- small math
// lo <= hi everywhere it is used: the median of (x, lo, hi) IS the clamp, in one v_med3_f32 instead of v_max + v_min
TE_DEV float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// gfx950 native transcendentals (1 ulp class, quarter-rate VALU) instead of the branchy libm slow paths:
// the sub-step loop is VALU-issue-bound, so instruction count is the lever.
TE_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }            // v_rcp_f32
TE_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }            // v_rsq_f32
TE_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }         // v_sqrt_f32
TE_DEV float sin_rev(float r) { return __builtin_amdgcn_sinf(r); }        // v_sin_f32: sin(2 pi r)
TE_DEV float cos_rev(float r) { return __builtin_amdgcn_cosf(r); }        // v_cos_f32: cos(2 pi r)
TE_DEV float ln(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }  // v_log_f32 is log2
TE_DEV float log2_native(float x) { return __builtin_amdgcn_logf(x); }
// atan on [0, inf) by one reduction around tan(pi/8) + a degree-4 odd minimax (Cephes atanf), ~1e-7 abs
TE_DEV float atan_poly(float x) {
  float z = x * x;
  float y = ((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f;
  return y * z * x + x;
}
TE_DEV float fast_atan2(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  // t = mn/mx in [0,1]; above tan(pi/8) use atan(t) = pi/4 + atan((t-1)/(t+1)) = pi/4 + atan((mn-mx)/(mn+mx))
  bool hi = mn > 0.4142135623730950f * mx;
  float num = hi ? mn - mx : mn, den = hi ? mn + mx : mx;
  float t = den > 0.0f ? num * rcp(den) : 0.0f;
  float a = atan_poly(t) + (hi ? 0.7853981633974483f : 0.0f);
  a = ay > ax ? 1.5707963267948966f - a : a;
  a = x < 0.0f ? kPi - a : a;
  return copysignf(a, y);
}
// asin on [-1,1] (Cephes asinf), ~1e-7 abs
TE_DEV float asin_poly(float z) {
  return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f);
}
TE_DEV float fast_asin(float x) {
  float a = fabsf(x);
  bool big = a > 0.5f;
  float z = big ? 0.5f * (1.0f - a) : a * a;
  float s = big ? fsqrt(z) : a;
  float r = s + s * z * asin_poly(z);
  r = big ? 1.5707963267948966f - 2.0f * r : r;
  return copysignf(r, x);
}
// ---- exact arithmetic: functions that several kernels inline and whose results are compared BITWISE between them (engage_kernel vs
// engage_slots_kernel, te_engage.hpp).  With -ffp-contract=fast the compiler chooses which multiply of a sum it fuses from the use counts
// around the expression, i.e. from the kernel it inlines into; TE_EXACT switches contraction off for the enclosing block and the fused
// multiply-adds are spelled out with xfma.
#define TE_EXACT _Pragma("clang fp contract(off)")
TE_DEV float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// sin and cos of a moderate x (|x| < ~1e4): Cody-Waite reduction by the nearest multiple of pi/2, Cephes sinf / cosf minimax polynomials
// (< 1 ulp of the result's scale); no libm, no branches
TE_DEV void x_sincos(float x, float& s, float& c) {
  TE_EXACT
  const float qf = rintf(x * 0.6366197723675814f);
  const int q = (int)qf;
  float r = xfma(qf, -1.5703125f, x);               // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188216e-8
  r = xfma(qf, -4.837512969970703125e-4f, r);
  r = xfma(qf, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float sr = xfma(r * z, xfma(xfma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), r);
  const float cr = xfma(z * z, xfma(xfma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), xfma(-0.5f, z, 1.0f));
  const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;   // sin(r + q pi/2): (s, c), (c, -s), (-s, -c), (-c, s)
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
// fast_asin / fast_atan2 above in Horner form on explicit fmas
TE_DEV float x_asin(float x) {
  TE_EXACT
  const float a = fabsf(x);
  const bool big = a > 0.5f;
  const float z = big ? 0.5f * (1.0f - a) : a * a;
  const float s = big ? fsqrt(z) : a;
  const float pz = xfma(xfma(xfma(xfma(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
  float r = xfma(s * z, pz, s);
  r = big ? xfma(-2.0f, r, 1.5707963267948966f) : r;
  return copysignf(r, x);
}
TE_DEV float x_atan2(float y, float x) {
  TE_EXACT
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const bool hi = mn > 0.4142135623730950f * mx;
  const float num = hi ? mn - mx : mn, den = hi ? mn + mx : mx;
  const float t = den > 0.0f ? num * rcp(den) : 0.0f;
  const float z = t * t;
  const float pz = xfma(xfma(xfma(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
  float a = xfma(pz * z, t, t) + (hi ? 0.7853981633974483f : 0.0f);
  a = ay > ax ? 1.5707963267948966f - a : a;
  a = x < 0.0f ? kPi - a : a;
  return copysignf(a, y);
}

struct V3 { float x, y, z; };
TE_DEV float norm(V3 a) { TE_EXACT return sqrtf(xfma(a.z, a.z, xfma(a.y, a.y, a.x * a.x))); }
TE_DEV V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
struct Q4 { float x, y, z, w; };
struct M3 { float m00, m01, m02, m10, m11, m12, m20, m21, m22; };

// rotation matrix of a quaternion (x,y,z,w), Bullet btMatrix3x3::setRotation form
TE_DEV M3 rotation(Q4 q) {
  float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  float s = 2.0f * rcp(d);
  float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  return M3{1.0f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0f - (xx + yy)};
}
// the same for a quaternion that is already normalised (the sub-step loop renormalises b.q every sub-step, as
// Bullet's integrator does): s = 2 / |q|^2 = 2 up to 1e-7, no reciprocal
TE_DEV M3 rotation_unit(Q4 q) {
  TE_EXACT
  const float xs = q.x + q.x, ys = q.y + q.y, zs = q.z + q.z;
  const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const float yy = q.y * ys, zz = q.z * zs;
  return M3{1.0f - xfma(q.y, ys, zz), xfma(q.x, ys, -wz), xfma(q.x, zs, wy), xfma(q.x, ys, wz), 1.0f - xfma(q.x, xs, zz), xfma(q.y, zs, -wx),
            xfma(q.x, zs, -wy), xfma(q.y, zs, wx), 1.0f - xfma(q.x, xs, yy)};
}
// rotation() in exact arithmetic (the flights' prologue and epilogue: inlined into every instantiation of the sub-step kernel)
TE_DEV M3 x_rotation(Q4 q) {
  TE_EXACT
  const float d = xfma(q.w, q.w, xfma(q.z, q.z, xfma(q.y, q.y, q.x * q.x)));
  const float s = 2.0f * rcp(d);
  const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const float yy = q.y * ys, zz = q.z * zs;
  return M3{1.0f - xfma(q.y, ys, zz), xfma(q.x, ys, -wz), xfma(q.x, zs, wy), xfma(q.x, ys, wz), 1.0f - xfma(q.x, xs, zz), xfma(q.y, zs, -wx),
            xfma(q.x, zs, -wy), xfma(q.y, zs, wx), 1.0f - xfma(q.x, xs, yy)};
}
TE_DEV V3 mul(const M3& R, V3 v) {
  TE_EXACT
  return V3{xfma(R.m02, v.z, xfma(R.m01, v.y, R.m00 * v.x)), xfma(R.m12, v.z, xfma(R.m11, v.y, R.m10 * v.x)), xfma(R.m22, v.z, xfma(R.m21, v.y, R.m20 * v.x))};
}
TE_DEV V3 mulT(const M3& R, V3 v) {
  TE_EXACT
  return V3{xfma(R.m20, v.z, xfma(R.m10, v.y, R.m00 * v.x)), xfma(R.m21, v.z, xfma(R.m11, v.y, R.m01 * v.x)), xfma(R.m22, v.z, xfma(R.m12, v.y, R.m02 * v.x))};
}
// roll/pitch/yaw with pybullet's gimbal guard (getEulerFromQuaternion); R entries equal the quaternion
// polynomials of that routine for a unit quaternion.  Polynomial atan2/asin (~1e-7): no libm in kernels.
TE_DEV V3 euler_of(Q4 q) {
  TE_EXACT
  const float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
  const float sarg = -2.0f * xfma(q.x, q.z, -(q.w * q.y));
  if (sarg <= -0.99999f) return V3{0.0f, -0.5f * kPi, 2.0f * x_atan2(q.x, -q.y)};
  if (sarg >= 0.99999f) return V3{0.0f, 0.5f * kPi, 2.0f * x_atan2(-q.x, q.y)};
  return V3{x_atan2(2.0f * xfma(q.y, q.z, q.w * q.x), ((sqw - sqx) - sqy) + sqz), x_asin(sarg),
            x_atan2(2.0f * xfma(q.x, q.y, q.w * q.z), ((sqw + sqx) - sqy) - sqz)};
}
// getQuaternionFromEuler (btQuaternion::setEulerZYX), normalised
TE_DEV Q4 quat_of_euler(V3 e) {
  float sr, cr, sp, cp, sy, cy;
  sincosf(0.5f * e.x, &sr, &cr); sincosf(0.5f * e.y, &sp, &cp); sincosf(0.5f * e.z, &sy, &cy);
  Q4 q{sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  float inv = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Q4{q.x * inv, q.y * inv, q.z * inv, q.w * inv};
}

// ---------------------------------------------------------------- quadrotor model
// Register-resident state of one drone during the sub-step loop.
struct Body {
  V3 pos; Q4 q; V3 vel; V3 wb;   // wb: angular velocity in BODY components (see integrate())
  float thr[4];
  float av_i[3], av_e[3], lv_i[2], lv_e[2], zv_i, zv_e;
  float pwm[4];  // QuadX.pwm between two controller updates: only live in the cfg.control_every_substep == 0 instantiation
  // IMU read (lagged observation)
  V3 o_pos, o_eul, o_vel, o_rate;
};

__host__ __device__ inline Derived derive(const te_config& c) {
  const te_quad_params& q = c.quad;
  Derived d;
  d.invT = 1.0f / c.control_dt; d.dt = c.physics_dt;
  d.k_motor = c.physics_dt / q.motor_tau; d.noise_ratio = q.noise_ratio;
  d.q_thrust = q.total_thrust * 0.25f;                      // thrust_coef * max_rpm^2
  d.q_arm = q.arm * d.q_thrust;
  d.q_torque = q.torque_coef * (q.total_thrust / (4.0f * q.thrust_coef));
  d.k_drag = 0.5f * q.air_density * q.drag_area_xyz * q.drag_coef_xyz; d.k_pqr = q.drag_coef_pqr;
  d.ix = q.inertia[0]; d.iy = q.inertia[1]; d.iz = q.inertia[2];
  d.dt_inv_ix = c.physics_dt / q.inertia[0]; d.dt_inv_iy = c.physics_dt / q.inertia[1]; d.dt_inv_iz = c.physics_dt / q.inertia[2];
  d.dt_inv_m = c.physics_dt / q.mass; d.dt_g = c.physics_dt * q.gravity; d.pwm_floor = q.pwm_floor;
  for (int i = 0; i < 2; ++i) { d.lv_kiT[i] = q.lin_vel_ki[i] * c.control_dt; d.lv_kdT[i] = q.lin_vel_kd[i] / c.control_dt; }
  for (int i = 0; i < 3; ++i) { d.av_kiT[i] = q.ang_vel_ki[i] * c.control_dt; d.av_kdT[i] = q.ang_vel_kd[i] / c.control_dt; }
  d.zv_kiT = q.z_vel_ki * c.control_dt; d.zv_kdT = q.z_vel_kd / c.control_dt;
  d.half_dt = 0.5f * c.physics_dt; d.quarter_dt2 = 0.25f * c.physics_dt * c.physics_dt;
  d.noise_m2ln2 = -2.0f * 0.69314718056f * q.noise_ratio * q.noise_ratio;
  d.ground_rest = c.ground_z + c.hull_half_height;
  d.ctrl_ratio = (int)(c.control_dt / c.physics_dt + 0.5f);
  if (d.ctrl_ratio < 1) d.ctrl_ratio = 1;
  return d;
}

// PyFlyt PID.step with the period folded into the gains (kiT = ki * T, kdT = kd / T): 2 fma + 2 clamps + 1 sub
TE_DEV float pid(float kp, float kiT, float kdT, float lim, float err, float& I, float& prev) {
  TE_EXACT
  I = clampf(fmaf(kiT, err, I), -lim, lim);
  const float d = err - prev;
  prev = err;
  return clampf(fmaf(kp, err, fmaf(kdT, d, I)), -lim, lim);
}

// Motor noise: 4 standard normals per sub-step.  One Philox4x32-7 call (14 quarter-rate 32x32->64 multiplies)
// serves TWO consecutive sub-steps: 128 bits = eight 16-bit uniforms = four Box-Muller pairs on the native
// log / sqrt / sin / cos.  Sub-step `sub` uses words {x,y} when even, {z,w} when odd, of call index sub >> 1.
// The two words travel into substep() as they are and become normals only at the motor stage, so that no
// float noise registers are live across the IMU / controller half of the sub-step.
TE_DEV U4 motor_noise_bits(const te_config& c, int env, int slot, uint32_t episode, uint32_t step_index, int sub) {
  // The counter and key of env_rng(c, env, RNG_MOTOR, slot, sub >> 1, episode, step_index), 7 rounds.  The empty asm statements make
  // the loop-invariant inputs opaque per call: otherwise the compiler hoists the invariant half of the first
  // Philox rounds and all 20 round keys out of the sub-step loop, into ~8 VGPRs and ~20 SGPRs that stay live
  // across it (97 -> 91 VGPRs, i.e. 4 -> 5 waves per SIMD, for three extra multiplies per call).
  uint32_t k0 = (uint32_t)c.seed, k1 = (uint32_t)(c.seed >> 32);
  asm volatile("" : "+s"(k0), "+s"(k1));
  asm volatile("" : "+v"(episode), "+v"(step_index), "+v"(env));
  const uint64_t g = (uint64_t)c.env_index_base + (uint64_t)env;
  return philox4x32<kMotorNoiseRounds>((uint32_t)g, RNG_MOTOR | ((uint32_t)slot << 8) | ((uint32_t)(sub >> 1) << 16) | ((uint32_t)(g >> 32) << 24),
                                       episode, step_index, k0, k1);
}
// `m2ln2_gain2` = -2 ln 2 * gain^2: the normals come out already multiplied by the noise gain.
TE_DEV void motor_noise_from(uint32_t a, uint32_t b, float m2ln2_gain2, float nz[4]) {
  TE_EXACT
  const float k16 = 1.0f / 65536.0f;
  float r0 = fsqrt(m2ln2_gain2 * log2_native(((float)(a & 0xFFFFu) + 0.5f) * k16)), r1 = fsqrt(m2ln2_gain2 * log2_native(((float)(b & 0xFFFFu) + 0.5f) * k16));
  float a0 = (float)(a >> 16) * k16, a1 = (float)(b >> 16) * k16;  // revolutions
  nz[0] = r0 * cos_rev(a0); nz[1] = r0 * sin_rev(a0); nz[2] = r1 * cos_rev(a1); nz[3] = r1 * sin_rev(a1);
}

// One physics sub-step: IMU read -> cascaded PID (mode 6, or 7 when MODE7) -> motors + drag -> free-body
// integration.  Replaces quadcopter.update_imu/update_control/update_physics + stepSimulation
// (level4_simulation.py:87-98) for one armed drone.  `sp` = [a0, a1, yaw-rate, z] set-point.
//
// Algebraic identities that keep the loop lean (exact in real arithmetic, ~1e-7 in float32):
//  * the body components of the angular velocity are invariant under the attitude update
//    (exp(w^ dt) w = w), so w stays in body axes for the whole loop and exp(dt w/2) is applied on the
//    right of q;
//  * cos/sin(yaw) come from the first column of R instead of sincos(atan2(.));
//  * sin/cos of the half rotation angle (< 0.5 rad per 1/240 s for any sane rate) by Taylor polynomials.
// CTRL: 1 = the controller runs in this sub-step and its pwm is used at once (the reference's loop: update_control on every physics
// sub-step, level4_simulation.py:92-94); for cfg.control_every_substep == 0 (PyFlyt's own 120 Hz): 0 = a controller sub-step whose
// pwm is also kept in b.pwm, 2 = a coasting sub-step that feeds the motors b.pwm again (no IMU angles, no PIDs, unless CAPTURE)
// EXACT ARITHMETIC (round 4): the function is inlined into every instantiation of the sub-step kernel (dense / mixed flights, noise on / off, both
// controller rates, three task families), whose results are compared bitwise (mixed = dense flights; shards = the whole range).  Contraction is
// off and every fused multiply-add is spelled out, so no instantiation can fuse a sum differently from another (te_device.hpp "exact arithmetic").
// HELP (small shards, te_env.hip: substeps_kernel<..., HELP>): the four normals of this sub-step come out of LDS, where the flight's sibling
// wave has left them (`nz_row`: this lane's 16 bytes of the sub-step's row) — the same motor_noise_bits / motor_noise_from values, computed by
// another wave.  The row is requested at the top of the sub-step and used at the motors; fly() has made sure that it is published.
template <bool MODE7, bool CAPTURE, bool NOISE, bool GROUND = false, int CTRL = 1, bool HELP = false>
TE_DEV void substep(const te_config& c, const Derived& k, Body& b, const float sp[4], uint32_t noise_a, uint32_t noise_b,
                    V3& pend_f, V3& pend_t, const volatile float4* nz_row = nullptr) {
  TE_EXACT
  const te_quad_params& qp = c.quad;
  const float dt = k.dt;
  float nzx = 0.0f, nzy = 0.0f, nzz = 0.0f, nzw = 0.0f;
  if (NOISE && HELP) { nzx = nz_row->x; nzy = nz_row->y; nzz = nz_row->z; nzw = nz_row->w; }
  M3 R = rotation_unit(b.q);
  // ---- IMU (imu.py:27-41)
  V3 vb = mulT(R, b.vel);
  float sarg = -R.m20;
  float roll = 0.0f, pitch = 0.0f, cyaw = 1.0f, syaw = 0.0f;
  bool guard = fabsf(sarg) >= 0.99999f;
  if (CTRL == 2 && !CAPTURE) {
    // coasting sub-step: nobody looks at the angles
  } else if (
#ifdef TE_NO_EASY_ANGLES   // A/B builds only (tools/ab.sh)
             false &&
#endif
             __builtin_amdgcn_ballot_w64(!(R.m22 > 0.0f && fabsf(R.m21) <= 0.4142135623730950f * R.m22 && fabsf(sarg) <= 0.5f)) == 0ull) {
    // Every lane of the wave flies within 22.5 degrees of roll and 30 degrees of pitch (the cascade's tilt limit is 0.4 rad): both
    // inverse functions are on their central branch, where x_atan2 / x_asin reduce to their odd polynomials — the same values, bit for bit,
    // without the range reduction, the square root and the selects of the general form (wave-uniform test; ~20 VALU instructions)
    const float t = R.m21 * rcp(R.m22);
    const float zt = t * t;
    const float pa = xfma(xfma(xfma(8.05374449538e-2f, zt, -1.38776856032e-1f), zt, 1.99777106478e-1f), zt, -3.33329491539e-1f);
    roll = xfma(pa * zt, t, t);
    const float zs = sarg * sarg;
    const float ps = xfma(xfma(xfma(xfma(4.2163199048e-2f, zs, 2.4181311049e-2f), zs, 4.5470025998e-2f), zs, 7.4953002686e-2f), zs, 1.6666752422e-1f);
    pitch = xfma(sarg * zs, ps, sarg);
    const float inv = rsq(xfma(R.m10, R.m10, R.m00 * R.m00));
    cyaw = R.m00 * inv; syaw = R.m10 * inv;
  } else if (!guard) {
    roll = x_atan2(R.m21, R.m22);
    pitch = x_asin(sarg);
    const float inv = rsq(xfma(R.m10, R.m10, R.m00 * R.m00));
    cyaw = R.m00 * inv; syaw = R.m10 * inv;
  } else {  // pybullet's gimbal guard (rare): roll = 0, pitch = +-pi/2, yaw = 2 atan2(+-x, -+y)
    roll = 0.0f; pitch = sarg > 0.0f ? 0.5f * kPi : -0.5f * kPi;
    float sy = sarg > 0.0f ? -b.q.x : b.q.x, cy = sarg > 0.0f ? b.q.y : -b.q.y;  // half-angle direction
    const float inv = rsq(fmaxf(xfma(cy, cy, sy * sy), 1e-30f));
    sy *= inv; cy *= inv;
    cyaw = xfma(cy, cy, -(sy * sy)); syaw = 2.0f * sy * cy;  // double angle
  }
  if (CAPTURE) {  // the IMU read the observation / reward / engagement will see (one-sub-step lag)
    b.o_pos = b.pos; b.o_vel = vb; b.o_rate = b.wb;
    b.o_eul = guard ? euler_of(b.q) : V3{roll, pitch, x_atan2(R.m10, R.m00)};
  }
  // ---- controller (PyFlyt QuadX.update_control; every sub-step, PID period control_dt)
  float pwm[4];
  if (CTRL == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pwm[i] = b.pwm[i];
  } else {
  float a0 = sp[0], a1 = sp[1], zc = sp[3];
  if (MODE7) {
    a0 = clampf(qp.lin_pos_kp[0] * (a0 - b.pos.x), -qp.lin_pos_lim[0], qp.lin_pos_lim[0]);
    a1 = clampf(qp.lin_pos_kp[1] * (a1 - b.pos.y), -qp.lin_pos_lim[1], qp.lin_pos_lim[1]);
    zc = clampf(qp.z_pos_kp * (zc - b.pos.z), -qp.z_pos_lim, qp.z_pos_lim);
  }
  const float u = xfma(syaw, a1, cyaw * a0), v = xfma(cyaw, a1, -(syaw * a0));
  float ox = pid(qp.lin_vel_kp[0], k.lv_kiT[0], k.lv_kdT[0], qp.lin_vel_lim[0], u - vb.x, b.lv_i[0], b.lv_e[0]);
  float oy = pid(qp.lin_vel_kp[1], k.lv_kiT[1], k.lv_kdT[1], qp.lin_vel_lim[1], v - vb.y, b.lv_i[1], b.lv_e[1]);
  float r0 = clampf(qp.ang_pos_kp[0] * (-oy - roll), -qp.ang_pos_lim[0], qp.ang_pos_lim[0]);
  float r1 = clampf(qp.ang_pos_kp[1] * (ox - pitch), -qp.ang_pos_lim[1], qp.ang_pos_lim[1]);
  float t0 = pid(qp.ang_vel_kp[0], k.av_kiT[0], k.av_kdT[0], qp.ang_vel_lim[0], r0 - b.wb.x, b.av_i[0], b.av_e[0]);
  float t1 = pid(qp.ang_vel_kp[1], k.av_kiT[1], k.av_kdT[1], qp.ang_vel_lim[1], r1 - b.wb.y, b.av_i[1], b.av_e[1]);
  float t2 = pid(qp.ang_vel_kp[2], k.av_kiT[2], k.av_kdT[2], qp.ang_vel_lim[2], sp[2] - b.wb.z, b.av_i[2], b.av_e[2]);
  float th = pid(qp.z_vel_kp, k.zv_kiT, k.zv_kdT, qp.z_vel_lim, zc - vb.z, b.zv_i, b.zv_e);
  th = clampf(th, 0.0f, 1.0f);
  pwm[0] = ((-t0 - t1) + t2) + th; pwm[1] = ((t0 + t1) + t2) + th; pwm[2] = ((-t0 + t1) - t2) + th; pwm[3] = ((t0 - t1) - t2) + th;
  float hi = fmaxf(fmaxf(pwm[0], pwm[1]), fmaxf(pwm[2], pwm[3]));
  if (__builtin_amdgcn_ballot_w64(hi > 1.0f) != 0ull && hi > 1.0f) {  // wave-uniform test first: saturation is rare
    float s = rcp(hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) pwm[i] *= s;
  }
  float lo = fminf(fminf(pwm[0], pwm[1]), fminf(pwm[2], pwm[3]));
  if (lo < k.pwm_floor) {
    float f = (k.pwm_floor - lo) * rcp(1.0f - lo);
#pragma unroll
    for (int i = 0; i < 4; ++i) pwm[i] = xfma(1.0f - pwm[i], f, pwm[i]);
  }
  if (CTRL == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) b.pwm[i] = pwm[i];
  }
  }
  // ---- motors (first-order lag, multiplicative noise, thrust/torque ~ rpm^2) + drag
  float T_[4], nz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (NOISE && HELP) { nz[0] = nzx; nz[1] = nzy; nz[2] = nzz; nz[3] = nzw; }
  else if (NOISE) motor_noise_from(noise_a, noise_b, k.noise_m2ln2, nz);  // nz = noise_ratio * N(0, 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = b.thr[i];
    t = xfma(k.k_motor, pwm[i] - t, t);
    if (NOISE) t = xfma(nz[i], t, t);
    b.thr[i] = t;
    T_[i] = t * t;
  }
  // layout m0 front-right (+x,-y), m1 back-left (-x,+y), m2 back-right (-x,-y), m3 front-left (+x,+y)
  const float fz = k.q_thrust * (((T_[0] + T_[1]) + T_[2]) + T_[3]);
  const float tx = k.q_arm * (((-T_[0] + T_[1]) - T_[2]) + T_[3]);
  const float ty = k.q_arm * (((-T_[0] + T_[1]) + T_[2]) - T_[3]);
  const float tz = k.q_torque * (((T_[0] + T_[1]) - T_[2]) - T_[3]);
  V3 Fb{(-k.k_drag * fabsf(vb.x)) * vb.x, (-k.k_drag * fabsf(vb.y)) * vb.y, xfma(-k.k_drag * fabsf(vb.z), vb.z, fz)};
  V3 Tb{xfma(-(k.k_pqr * fabsf(b.wb.x)), b.wb.x, tx), xfma(-(k.k_pqr * fabsf(b.wb.y)), b.wb.y, ty), xfma(-(k.k_pqr * fabsf(b.wb.z)), b.wb.z, tz)};
  V3 Fw = mul(R, Fb);
  if (MODE7) {  // wrench accumulated outside the loop (stage01 replace_invader), world frame
    Fw.x += pend_f.x; Fw.y += pend_f.y; Fw.z += pend_f.z;
    V3 tb = mulT(R, pend_t);
    Tb.x += tb.x; Tb.y += tb.y; Tb.z += tb.z;
    pend_f = V3{0, 0, 0}; pend_t = V3{0, 0, 0};
  }
  // ---- Bullet semi-implicit Euler (stepSimulation): gyroscopic term, exponential-map attitude update
  const V3 Iw{k.ix * b.wb.x, k.iy * b.wb.y, k.iz * b.wb.z};
  const V3 gy{xfma(b.wb.y, Iw.z, -(b.wb.z * Iw.y)), xfma(b.wb.z, Iw.x, -(b.wb.x * Iw.z)), xfma(b.wb.x, Iw.y, -(b.wb.y * Iw.x))};
  b.wb = V3{xfma(k.dt_inv_ix, Tb.x - gy.x, b.wb.x), xfma(k.dt_inv_iy, Tb.y - gy.y, b.wb.y), xfma(k.dt_inv_iz, Tb.z - gy.z, b.wb.z)};
  b.vel = V3{xfma(k.dt_inv_m, Fw.x, b.vel.x), xfma(k.dt_inv_m, Fw.y, b.vel.y), b.vel.z + xfma(k.dt_inv_m, Fw.z, -k.dt_g)};
  b.pos = V3{xfma(dt, b.vel.x, b.pos.x), xfma(dt, b.vel.y, b.pos.y), xfma(dt, b.vel.z, b.pos.z)};
  if (GROUND && c.ground_contact && b.pos.z < k.ground_rest) {  // opt-in ground plane (config-uniform test first): inelastic normal
    // contact, Coulomb friction mu = 0.5 (Bullet's default 0.5 x plane.urdf's 1.0) against the normal impulse; unpinned
    const float jn = fmaxf(-b.vel.z, 0.0f);
    b.pos.z = k.ground_rest; b.vel.z = fmaxf(b.vel.z, 0.0f);
    const float vt2 = xfma(b.vel.y, b.vel.y, b.vel.x * b.vel.x);
    if (vt2 > 0.0f) {
      const float vt = fsqrt(vt2), keep = fmaxf(xfma(-0.5f, jn, vt), 0.0f) * rcp(vt);
      b.vel.x *= keep; b.vel.y *= keep;
    }
  }
  const float w2 = xfma(b.wb.z, b.wb.z, xfma(b.wb.y, b.wb.y, b.wb.x * b.wb.x));
  const float h2 = k.quarter_dt2 * w2;  // (half angle)^2
  float sc, ch;                     // sin(h)/|w| and cos(h)
  if (h2 < 0.25f) {
    sc = k.half_dt * xfma(h2, xfma(h2, xfma(h2, xfma(h2, 1.0f / 362880.0f, -1.0f / 5040.0f), 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
    ch = xfma(h2, xfma(h2, xfma(h2, xfma(h2, xfma(h2, -1.0f / 3628800.0f, 1.0f / 40320.0f), -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
  } else {  // > 240 rad/s: native sin/cos (inputs in revolutions) are plenty
    const float inv_w = rsq(w2);
    const float rev = (k.half_dt * (w2 * inv_w)) * (0.5f / kPi);
    sc = sin_rev(rev) * inv_w; ch = cos_rev(rev);
  }
  const Q4 dq{b.wb.x * sc, b.wb.y * sc, b.wb.z * sc, ch};
  const Q4 q = b.q;  // q <- q (x) dq   (body-frame increment on the right)
  const Q4 n{xfma(-q.z, dq.y, xfma(q.y, dq.z, xfma(q.x, dq.w, q.w * dq.x))), xfma(q.z, dq.x, xfma(q.y, dq.w, xfma(-q.x, dq.z, q.w * dq.y))),
             xfma(q.z, dq.w, xfma(-q.y, dq.x, xfma(q.x, dq.y, q.w * dq.z))), xfma(-q.z, dq.z, xfma(-q.y, dq.y, xfma(-q.x, dq.x, q.w * dq.w)))};
  const float inv = rsq(xfma(n.w, n.w, xfma(n.z, n.z, xfma(n.y, n.y, n.x * n.x))));
  b.q = Q4{n.x * inv, n.y * inv, n.z * inv, n.w * inv};
}

// Quadcopter.convert_command_to_setpoint (quadcopter.py:379-396): unit(direction) * magnitude
TE_DEV void command_to_velocity(float dx, float dy, float dz, float mag, float& vx, float& vy, float& vz) {
  TE_EXACT
  float n = sqrtf(xfma(dz, dz, xfma(dy, dy, dx * dx)));
  float inv = 1.0f / (n > 0.0f ? n : 1.0f);
  vx = mag * (dx * inv); vy = mag * (dy * inv); vz = mag * (dz * inv);
}

}  // namespace te
