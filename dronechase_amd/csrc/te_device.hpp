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
enum { TE_X_CMD = TE_DRONE_WORDS, /* 3: next scripted velocity command vx,vy,vz */
       TE_X_NAV_NEXT = TE_DRONE_WORDS + 3, /* i32: FSM state after the pending update */
       TE_X_WORDS = 4 };

struct Params {
  te_config cfg;
  uint32_t* dstate;
  uint32_t* estate;
  int N, Npad, D;
};

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

// ---------------------------------------------------------------- Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
TE_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
enum { RNG_SPAWN_INVADER = 1, RNG_SPAWN_PURSUER = 2, RNG_HIT = 3, RNG_MOTOR = 4, RNG_ACTION = 5, RNG_RESPAWN = 6 };
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
TE_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
struct V3 { float x, y, z; };
TE_DEV float norm(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
TE_DEV V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
struct Q4 { float x, y, z, w; };
struct M3 { float m00, m01, m02, m10, m11, m12, m20, m21, m22; };

// rotation matrix of a quaternion (x,y,z,w), Bullet btMatrix3x3::setRotation form
TE_DEV M3 rotation(Q4 q) {
  float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  float s = 2.0f / d;
  float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  return M3{1.0f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0f - (xx + yy)};
}
TE_DEV V3 mul(const M3& R, V3 v) {
  return V3{R.m00 * v.x + R.m01 * v.y + R.m02 * v.z, R.m10 * v.x + R.m11 * v.y + R.m12 * v.z,
            R.m20 * v.x + R.m21 * v.y + R.m22 * v.z};
}
TE_DEV V3 mulT(const M3& R, V3 v) {
  return V3{R.m00 * v.x + R.m10 * v.y + R.m20 * v.z, R.m01 * v.x + R.m11 * v.y + R.m21 * v.z,
            R.m02 * v.x + R.m12 * v.y + R.m22 * v.z};
}
// roll/pitch/yaw with pybullet's gimbal guard (getEulerFromQuaternion); R entries equal the quaternion
// polynomials of that routine for a unit quaternion.
TE_DEV V3 euler_of(Q4 q) {
  float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
  float sarg = -2.0f * (q.x * q.z - q.w * q.y);
  if (sarg <= -0.99999f) return V3{0.0f, -0.5f * kPi, 2.0f * atan2f(q.x, -q.y)};
  if (sarg >= 0.99999f) return V3{0.0f, 0.5f * kPi, 2.0f * atan2f(-q.x, q.y)};
  return V3{atan2f(2.0f * (q.y * q.z + q.w * q.x), sqw - sqx - sqy + sqz), asinf(sarg),
            atan2f(2.0f * (q.x * q.y + q.w * q.z), sqw + sqx - sqy - sqz)};
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
  // IMU read (lagged observation)
  V3 o_pos, o_eul, o_vel, o_rate;
};

// PyFlyt PID.step
TE_DEV float pid(float kp, float ki, float kd, float lim, float T, float err, float& I, float& prev) {
  I = clampf(I + ki * err * T, -lim, lim);
  float Dv = kd * (err - prev) / T;
  prev = err;
  return clampf(kp * err + I + Dv, -lim, lim);
}

// One physics sub-step: IMU read -> cascaded PID (mode 6, or 7 when MODE7) -> motors + drag -> free-body
// integration.  Replaces quadcopter.update_imu/update_control/update_physics + stepSimulation
// (level4_simulation.py:87-98) for one armed drone.  `sp` = [a0, a1, yaw-rate, z] set-point.
//
// Two algebraic identities keep the loop lean (both exact in real arithmetic):
//  * the body components of the angular velocity are invariant under the attitude update
//    (exp(w^ dt) w = w), so w stays in body axes for the whole loop and exp(dt w/2) is applied on the
//    right of q;
//  * cos/sin(yaw) come from the first column of R instead of sincos(atan2(.)).
template <bool MODE7>
TE_DEV void substep(const te_config& c, Body& b, const float sp[4], const float nz[4], V3& pend_f, V3& pend_t,
                    bool last) {
  const te_quad_params& qp = c.quad;
  const float T = c.control_dt, dt = c.physics_dt;
  M3 R = rotation(b.q);
  // ---- IMU (imu.py:27-41)
  V3 vb = mulT(R, b.vel);
  float sarg = -R.m20;
  float roll, pitch, cyaw, syaw;
  bool guard = fabsf(sarg) >= 0.99999f;
  if (!guard) {
    roll = atan2f(R.m21, R.m22);
    pitch = asinf(sarg);
    float inv = 1.0f / sqrtf(R.m00 * R.m00 + R.m10 * R.m10);
    cyaw = R.m00 * inv; syaw = R.m10 * inv;
  } else {
    V3 e = euler_of(b.q);
    roll = e.x; pitch = e.y; sincosf(e.z, &syaw, &cyaw);
  }
  if (last) {
    b.o_pos = b.pos; b.o_vel = vb; b.o_rate = b.wb;
    b.o_eul = guard ? euler_of(b.q) : V3{roll, pitch, atan2f(R.m10, R.m00)};
  }
  // ---- controller (PyFlyt QuadX.update_control; every sub-step, PID period control_dt)
  float a0 = sp[0], a1 = sp[1], zc = sp[3];
  if (MODE7) {
    a0 = clampf(qp.lin_pos_kp[0] * (a0 - b.pos.x), -qp.lin_pos_lim[0], qp.lin_pos_lim[0]);
    a1 = clampf(qp.lin_pos_kp[1] * (a1 - b.pos.y), -qp.lin_pos_lim[1], qp.lin_pos_lim[1]);
    zc = clampf(qp.z_pos_kp * (zc - b.pos.z), -qp.z_pos_lim, qp.z_pos_lim);
  }
  float u = cyaw * a0 + syaw * a1, v = -syaw * a0 + cyaw * a1;
  float ox = pid(qp.lin_vel_kp[0], qp.lin_vel_ki[0], qp.lin_vel_kd[0], qp.lin_vel_lim[0], T, u - vb.x, b.lv_i[0], b.lv_e[0]);
  float oy = pid(qp.lin_vel_kp[1], qp.lin_vel_ki[1], qp.lin_vel_kd[1], qp.lin_vel_lim[1], T, v - vb.y, b.lv_i[1], b.lv_e[1]);
  float r0 = clampf(qp.ang_pos_kp[0] * (-oy - roll), -qp.ang_pos_lim[0], qp.ang_pos_lim[0]);
  float r1 = clampf(qp.ang_pos_kp[1] * (ox - pitch), -qp.ang_pos_lim[1], qp.ang_pos_lim[1]);
  float t0 = pid(qp.ang_vel_kp[0], qp.ang_vel_ki[0], qp.ang_vel_kd[0], qp.ang_vel_lim[0], T, r0 - b.wb.x, b.av_i[0], b.av_e[0]);
  float t1 = pid(qp.ang_vel_kp[1], qp.ang_vel_ki[1], qp.ang_vel_kd[1], qp.ang_vel_lim[1], T, r1 - b.wb.y, b.av_i[1], b.av_e[1]);
  float t2 = pid(qp.ang_vel_kp[2], qp.ang_vel_ki[2], qp.ang_vel_kd[2], qp.ang_vel_lim[2], T, sp[2] - b.wb.z, b.av_i[2], b.av_e[2]);
  float th = pid(qp.z_vel_kp, qp.z_vel_ki, qp.z_vel_kd, qp.z_vel_lim, T, zc - vb.z, b.zv_i, b.zv_e);
  th = clampf(th, 0.0f, 1.0f);
  float pwm[4] = {-t0 - t1 + t2 + th, t0 + t1 + t2 + th, -t0 + t1 - t2 + th, t0 - t1 - t2 + th};
  float hi = fmaxf(fmaxf(pwm[0], pwm[1]), fmaxf(pwm[2], pwm[3]));
  if (hi > 1.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pwm[i] /= hi;
  }
  float lo = fminf(fminf(pwm[0], pwm[1]), fminf(pwm[2], pwm[3]));
  if (lo < qp.pwm_floor) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pwm[i] += (1.0f - pwm[i]) / (1.0f - lo) * (qp.pwm_floor - lo);
  }
  // ---- motors (first-order lag, multiplicative noise, thrust/torque ~ rpm^2) + drag
  const float k = dt / qp.motor_tau;
  const float max_rpm2 = qp.total_thrust / (4.0f * qp.thrust_coef);
  float T_[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = b.thr[i];
    t += k * (pwm[i] - t);
    t += nz[i] * t * qp.noise_ratio;
    b.thr[i] = t;
    T_[i] = t * t * max_rpm2;
  }
  // layout m0 front-right (+x,-y), m1 back-left (-x,+y), m2 back-right (-x,-y), m3 front-left (+x,+y)
  float fz = qp.thrust_coef * (T_[0] + T_[1] + T_[2] + T_[3]);
  float tx = qp.arm * qp.thrust_coef * (-T_[0] + T_[1] - T_[2] + T_[3]);
  float ty = qp.arm * qp.thrust_coef * (-T_[0] + T_[1] + T_[2] - T_[3]);
  float tz = qp.torque_coef * (T_[0] + T_[1] - T_[2] - T_[3]);
  const float kd = 0.5f * qp.air_density * qp.drag_area_xyz * qp.drag_coef_xyz;
  V3 Fb{-kd * fabsf(vb.x) * vb.x, -kd * fabsf(vb.y) * vb.y, -kd * fabsf(vb.z) * vb.z + fz};
  V3 Tb{tx - qp.drag_coef_pqr * fabsf(b.wb.x) * b.wb.x, ty - qp.drag_coef_pqr * fabsf(b.wb.y) * b.wb.y,
        tz - qp.drag_coef_pqr * fabsf(b.wb.z) * b.wb.z};
  V3 Fw = mul(R, Fb);
  if (MODE7) {  // wrench accumulated outside the loop (stage01 replace_invader), world frame
    Fw.x += pend_f.x; Fw.y += pend_f.y; Fw.z += pend_f.z;
    V3 tb = mulT(R, pend_t);
    Tb.x += tb.x; Tb.y += tb.y; Tb.z += tb.z;
    pend_f = V3{0, 0, 0}; pend_t = V3{0, 0, 0};
  }
  // ---- Bullet semi-implicit Euler (stepSimulation): gyroscopic term, exponential-map attitude update
  float Ix = qp.inertia[0], Iy = qp.inertia[1], Iz = qp.inertia[2];
  V3 Iw{Ix * b.wb.x, Iy * b.wb.y, Iz * b.wb.z};
  V3 gy{b.wb.y * Iw.z - b.wb.z * Iw.y, b.wb.z * Iw.x - b.wb.x * Iw.z, b.wb.x * Iw.y - b.wb.y * Iw.x};
  b.wb = V3{b.wb.x + dt * (Tb.x - gy.x) / Ix, b.wb.y + dt * (Tb.y - gy.y) / Iy, b.wb.z + dt * (Tb.z - gy.z) / Iz};
  float inv_m = 1.0f / qp.mass;
  b.vel = V3{b.vel.x + dt * Fw.x * inv_m, b.vel.y + dt * Fw.y * inv_m, b.vel.z + dt * (Fw.z * inv_m - qp.gravity)};
  b.pos = V3{b.pos.x + dt * b.vel.x, b.pos.y + dt * b.vel.y, b.pos.z + dt * b.vel.z};
  float wmag = norm(b.wb);
  float half = 0.5f * wmag * dt;
  float sc, ch;
  if (wmag < 1e-6f) { sc = 0.5f * dt * (1.0f - half * half / 6.0f); ch = cosf(half); }
  else { float sh; sincosf(half, &sh, &ch); sc = sh / wmag; }
  Q4 dq{b.wb.x * sc, b.wb.y * sc, b.wb.z * sc, ch};
  Q4 q = b.q;  // q <- q (x) dq   (body-frame increment on the right)
  Q4 n{q.w * dq.x + q.x * dq.w + q.y * dq.z - q.z * dq.y, q.w * dq.y - q.x * dq.z + q.y * dq.w + q.z * dq.x,
       q.w * dq.z + q.x * dq.y - q.y * dq.x + q.z * dq.w, q.w * dq.w - q.x * dq.x - q.y * dq.y - q.z * dq.z};
  float inv = 1.0f / sqrtf(n.x * n.x + n.y * n.y + n.z * n.z + n.w * n.w);
  b.q = Q4{n.x * inv, n.y * inv, n.z * inv, n.w * inv};
}

TE_DEV void motor_noise(const te_config& c, int env, int slot, uint32_t episode, uint32_t step_index, int sub, float nz[4]) {
  U4 r = env_rng(c, env, RNG_MOTOR, (uint32_t)slot, (uint32_t)sub, episode, step_index);
  float r0 = sqrtf(-2.0f * logf(u01_open(r.x))), r1 = sqrtf(-2.0f * logf(u01_open(r.z)));
  float s0, c0, s1, c1;
  sincosf(2.0f * kPi * u01(r.y), &s0, &c0);
  sincosf(2.0f * kPi * u01(r.w), &s1, &c1);
  nz[0] = r0 * c0; nz[1] = r0 * s0; nz[2] = r1 * c1; nz[3] = r1 * s1;
}

// Quadcopter.convert_command_to_setpoint (quadcopter.py:379-396): unit(direction) * magnitude
TE_DEV void command_to_velocity(float dx, float dy, float dz, float mag, float& vx, float& vy, float& vz) {
  float n = sqrtf(dx * dx + dy * dy + dz * dz);
  float inv = 1.0f / (n > 0.0f ? n : 1.0f);
  vx = mag * (dx * inv); vy = mag * (dy * inv); vz = mag * (dz * inv);
}

}  // namespace te
