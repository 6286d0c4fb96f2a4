/*
 * te_oracle.c — TEST INFRASTRUCTURE ONLY.  Scalar CPU restatement of the reference's
 * env.step() hot path (DaviGuanabara/dronechase), used as the parity checker for the HIP
 * kernels and as the "port" CPU baseline of bench.py.  Nothing under dronechase_amd/ may
 * import, link or call this file (see DESIGN.md "oracle").
 *
 * Pinning status
 *   - task logic, LIDAR maths, gun, navigators, normalisation, cone geometry: pinned by golden
 *     vectors generated from the reference's own importable modules (tests/golden/gen_golden.py)
 *     and by the known-answer scenarios of the reference's tests.
 *   - quadrotor physics (PyFlyt 0.11.1 QuadX + pybullet 3.2.7, both absent from the reference tree
 *     and from this container): restated from their published algorithms;
 *     PARITY UNPINNED against PyBullet itself.
 *
 * All citations are file:line under /root/reference/src unless noted.
 * Build: see oracle/Makefile (twice: -DOTE_REAL=float and -DOTE_REAL=double).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tgmath.h>
#undef I

#include "../include/threatengage.h"

#ifndef OTE_REAL
#define OTE_REAL double
#endif
typedef OTE_REAL real;

#define OTE_PI ((real)3.14159265358979323846)
#define OTE_MAX_DRONES 64

#if defined(__GNUC__)
#define OTE_API __attribute__((visibility("default")))
#else
#define OTE_API
#endif

/* ------------------------------------------------------------------------- */
/* records                                                                    */
/* ------------------------------------------------------------------------- */
typedef struct {
  real pos[3], quat[4], vel[3], omega[3];
  real throttle[4];
  real av_i[3], av_e[3], lv_i[2], lv_e[2], zv_i, zv_e;
  real setpoint[4];
  real obs_pos[3], obs_euler[3], obs_vel[3], obs_rate[3];
  real formation[3];
  real pending[6];
  real pwm[4]; /* QuadX.pwm: the last update_control's output, what update_physics feeds the motors (only read back when
                  cfg.control_every_substep == 0; every env.step starts with a control update, so it is not part of the blob) */
  real ext_action[4]; /* caller-driven pursuer: the driver's last action (exp05_vFinal_task.py:139,259; evaluation_task.py:266); blob: TE_D_ALLY_ACTION */
  int32_t armed, munition, last_fired, nav_state;
} ote_drone;

typedef struct {
  int32_t step, max_step, round, info_wave;
  real last_dist;
  int32_t agent_kills, allies_kills, deads;
  uint64_t snap_mask; /* bit s = drone s armed when the offsets were last computed (blob: TE_E_SNAP_MASK + TE_E_SNAP_MASK_HI) */
  int32_t episode;
  real last_action[4];
  real prev_snap_min;
} ote_envrec;

typedef struct ote_env {
  te_config cfg;
  int D;
  ote_drone* drones; /* [N*D] */
  ote_envrec* envs;  /* [N] */
  real* margin;      /* [N] min |value - threshold| over ALL discrete decisions of the last step */
  real* margin_state; /* [N] same, restricted to decisions that change state or done (not reward-only ones) */
  /* level5 (cfg.stacked_obs): per-wingman snapshot ring, same word layout as the product's state blob
   * (include/threatengage.h TE_RING_*): ring[((e * P + p) * TE_RING_DEPTH + r) * entry_words] */
  uint32_t* ring;
  int entry_words;
  real* margin_stack; /* [N] smallest angular distance (rad) of any binned feature to a LIDAR cell boundary, last step */
  /* outputs of the step in flight (set by ote_step_stacked, NULL otherwise) */
  float* out_stacked; uint8_t* out_mask; float* out_t_stacked; uint8_t* out_t_mask;
  /* ote_step_students: [N,P,...] outputs of the step in flight */
  float* st_stacked; uint8_t* st_mask; float* st_inertial; float* st_last_action; uint8_t* st_active;
} ote_env;

/* ------------------------------------------------------------------------- */
/* counter-based RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 reference vectors are     */
/* checked in tests/test_oracle_units.py).  The reference uses unseeded global numpy/random     */
/* streams (exp03_vFinal_task.py:588-600, gun.py:94), so sequences cannot be reproduced; draws  */
/* are keyed (seed; global env, purpose, episode, sub-counter) instead.                         */
/* ------------------------------------------------------------------------- */
enum { OTE_RNG_SPAWN_INVADER = 1, OTE_RNG_SPAWN_PURSUER = 2, OTE_RNG_HIT = 3, OTE_RNG_MOTOR = 4,
       OTE_RNG_ACTION = 5, OTE_RNG_RESPAWN = 6, OTE_RNG_STACK = 7 };

/* Philox4x32-R (Salmon et al., SC'11; Random123): R = 10 for every draw that decides something, R = 7 (the smallest Crush-resistant
 * round count of the paper; Random123 ships known answers for it) for the motor noise */
static void philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_r(ctr, key, out, 10); }
OTE_API void ote_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  philox4x32_10(ctr, key, out);
}
OTE_API void ote_philox4x32_7(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_r(ctr, key, out, 7); }

/* uniform in [0,1): 24 high bits, exact in float and double */
static real u01(uint32_t x) { return (real)(x >> 8) * (real)(1.0 / 16777216.0); }



/* counter = { global env (low 32), purpose | slot<<8 | sub<<16 | global env (high 8)<<24, episode, index } */
static void ote_rng(const ote_env* E, int env_local, uint32_t purpose, uint32_t slot, uint32_t sub, uint32_t episode,
                    uint32_t index, uint32_t out[4]) {
  uint64_t g = (uint64_t)E->cfg.env_index_base + (uint64_t)env_local;
  uint32_t ctr[4] = {(uint32_t)g, purpose | (slot << 8) | (sub << 16) | ((uint32_t)(g >> 32) << 24), episode, index};
  uint32_t key[2] = {(uint32_t)E->cfg.seed, (uint32_t)(E->cfg.seed >> 32)};
  philox4x32_10(ctr, key, out);
}

/* ------------------------------------------------------------------------- */
/* Bullet math helpers (pybullet 3.2.7 C API, restated; call sites imu.py:33-38,                */
/* lidar_math.py:75,81,179, quadcopter.py:434)                                                  */
/* ------------------------------------------------------------------------- */
static real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
static real norm3(const real v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* getMatrixFromQuaternion: btMatrix3x3::setRotation, row-major, q = (x,y,z,w) */
static void quat_to_mat(const real q[4], real m[9]) {
  real d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  real s = (real)2 / d;
  real xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  real wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  real xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  real yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  m[0] = (real)1 - (yy + zz); m[1] = xy - wz;             m[2] = xz + wy;
  m[3] = xy + wz;             m[4] = (real)1 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy;             m[7] = yz + wx;             m[8] = (real)1 - (xx + yy);
}
static void mat_vec(const real m[9], const real v[3], real o[3]) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matT_vec(const real m[9], const real v[3], real o[3]) {
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
/* rotateVector(q, v) = q v q^-1 */
static void rotate_vector(const real q[4], const real v[3], real o[3]) {
  real m[9];
  quat_to_mat(q, m);
  mat_vec(m, v, o);
}
/* getEulerFromQuaternion (pybullet.c): roll, pitch, yaw with the +-pi/2 pitch guard */
static void euler_from_quat(const real q[4], real rpy[3]) {
  real sqx = q[0] * q[0], sqy = q[1] * q[1], sqz = q[2] * q[2], squ = q[3] * q[3];
  real sarg = (real)-2 * (q[0] * q[2] - q[3] * q[1]);
  if (sarg <= (real)-0.99999) {
    rpy[0] = 0; rpy[1] = (real)-0.5 * OTE_PI; rpy[2] = (real)2 * atan2(q[0], -q[1]);
  } else if (sarg >= (real)0.99999) {
    rpy[0] = 0; rpy[1] = (real)0.5 * OTE_PI; rpy[2] = (real)2 * atan2(-q[0], q[1]);
  } else {
    rpy[0] = atan2((real)2 * (q[1] * q[2] + q[3] * q[0]), squ - sqx - sqy + sqz);
    rpy[1] = asin(sarg);
    rpy[2] = atan2((real)2 * (q[0] * q[1] + q[3] * q[2]), squ + sqx - sqy - sqz);
  }
}
/* getQuaternionFromEuler: btQuaternion::setEulerZYX(yaw, pitch, roll), normalised */
static void quat_from_euler(const real rpy[3], real q[4]) {
  real hr = rpy[0] * (real)0.5, hp = rpy[1] * (real)0.5, hy = rpy[2] * (real)0.5;
  real cr = cos(hr), sr = sin(hr), cp = cos(hp), sp = sin(hp), cy = cos(hy), sy = sin(hy);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

OTE_API void ote_quat_to_mat(const double* q, double* m) {
  real qq[4] = {(real)q[0], (real)q[1], (real)q[2], (real)q[3]}, mm[9];
  quat_to_mat(qq, mm);
  for (int i = 0; i < 9; ++i) m[i] = (double)mm[i];
}
OTE_API void ote_euler_from_quat(const double* q, double* rpy) {
  real qq[4] = {(real)q[0], (real)q[1], (real)q[2], (real)q[3]}, e[3];
  euler_from_quat(qq, e);
  for (int i = 0; i < 3; ++i) rpy[i] = (double)e[i];
}
OTE_API void ote_quat_from_euler(const double* rpy, double* q) {
  real e[3] = {(real)rpy[0], (real)rpy[1], (real)rpy[2]}, qq[4];
  quat_from_euler(e, qq);
  for (int i = 0; i < 4; ++i) q[i] = (double)qq[i];
}
OTE_API void ote_rotate_vector(const double* q, const double* v, double* o) {
  real qq[4] = {(real)q[0], (real)q[1], (real)q[2], (real)q[3]};
  real vv[3] = {(real)v[0], (real)v[1], (real)v[2]}, oo[3];
  rotate_vector(qq, vv, oo);
  for (int i = 0; i < 3; ++i) o[i] = (double)oo[i];
}

/* ------------------------------------------------------------------------- */
/* L0: PyFlyt QuadX (cf2x) controller + motors + drag, Bullet free-body integration             */
/* (call sites quadcopter.py:543-549,557-566)                              */
/* ------------------------------------------------------------------------- */

/* PyFlyt PID.step: I = clip(I + ki e T); D = kd (e - e_prev)/T; out = clip(kp e + I + D) */
static real pid_step(real kp, real ki, real kd, real lim, real T, real err, real* integ, real* prev) {
  real I = clampr(*integ + ki * err * T, -lim, lim);
  real Dv = kd * (err - *prev) / T;
  *integ = I;
  *prev = err;
  return clampr(kp * err + I + Dv, -lim, lim);
}

/* IMU read: QuadX.update_state via InertialMeasurementUnit.update_data (imu.py:27-41):
 * body-frame linear/angular velocity (rotation^T . world), euler from quaternion, position. */
static void observe(ote_drone* d) {
  real m[9];
  quat_to_mat(d->quat, m);
  matT_vec(m, d->vel, d->obs_vel);
  matT_vec(m, d->omega, d->obs_rate);
  euler_from_quat(d->quat, d->obs_euler);
  d->obs_pos[0] = d->pos[0]; d->obs_pos[1] = d->pos[1]; d->obs_pos[2] = d->pos[2];
}

/* QuadX.update_control, flight mode 6 (vx, vy, vr, vz; quadcopter.py:152,408-413) or mode 7
 * (x, y, r, z; level2/components/quadcopter_manager.py:68).  Runs on EVERY physics sub-step
 * with the PID period still control_dt (level4_simulation.py:92-94). */
static void control(const te_config* c, ote_drone* d, int mode, real pwm[4]) {
  const te_quad_params* q = &c->quad;
  const real T = (real)c->control_dt;
  real a0 = d->setpoint[0], a1 = d->setpoint[1], a2 = d->setpoint[2], z = d->setpoint[3];
  if (mode == 7) {
    /* position loops (ki = kd = 0 in cf2x): world-frame velocity command */
    a0 = clampr((real)q->lin_pos_kp[0] * (a0 - d->obs_pos[0]), -(real)q->lin_pos_lim[0], (real)q->lin_pos_lim[0]);
    a1 = clampr((real)q->lin_pos_kp[1] * (a1 - d->obs_pos[1]), -(real)q->lin_pos_lim[1], (real)q->lin_pos_lim[1]);
    z = clampr((real)q->z_pos_kp * (z - d->obs_pos[2]), -(real)q->z_pos_lim, (real)q->z_pos_lim);
  }
  /* ground-frame velocity set-point -> body yaw frame */
  real cy = cos(d->obs_euler[2]), sy = sin(d->obs_euler[2]);
  real u = cy * a0 + sy * a1;
  real v = -sy * a0 + cy * a1;
  /* linear velocity -> desired tilt; swap (-out_y, out_x) = (roll, pitch) */
  real ox = pid_step((real)q->lin_vel_kp[0], (real)q->lin_vel_ki[0], (real)q->lin_vel_kd[0], (real)q->lin_vel_lim[0], T,
                     u - d->obs_vel[0], &d->lv_i[0], &d->lv_e[0]);
  real oy = pid_step((real)q->lin_vel_kp[1], (real)q->lin_vel_ki[1], (real)q->lin_vel_kd[1], (real)q->lin_vel_lim[1], T,
                     v - d->obs_vel[1], &d->lv_i[1], &d->lv_e[1]);
  real roll_des = -oy, pitch_des = ox;
  /* angular position -> angular rate (memory-less: ki = kd = 0); yaw channel is the rate vr itself */
  real r0 = clampr((real)q->ang_pos_kp[0] * (roll_des - d->obs_euler[0]), -(real)q->ang_pos_lim[0], (real)q->ang_pos_lim[0]);
  real r1 = clampr((real)q->ang_pos_kp[1] * (pitch_des - d->obs_euler[1]), -(real)q->ang_pos_lim[1], (real)q->ang_pos_lim[1]);
  real r2 = a2;
  /* angular rate -> normalised torque */
  real rates[3] = {r0, r1, r2}, tq[3];
  for (int i = 0; i < 3; ++i)
    tq[i] = pid_step((real)q->ang_vel_kp[i], (real)q->ang_vel_ki[i], (real)q->ang_vel_kd[i], (real)q->ang_vel_lim[i], T,
                     rates[i] - d->obs_rate[i], &d->av_i[i], &d->av_e[i]);
  /* vertical velocity -> normalised thrust */
  real th = pid_step((real)q->z_vel_kp, (real)q->z_vel_ki, (real)q->z_vel_kd, (real)q->z_vel_lim, T,
                     z - d->obs_vel[2], &d->zv_i, &d->zv_e);
  th = clampr(th, (real)0, (real)1);
  /* motor mix (X configuration, rows = motors, cols = roll, pitch, yaw, thrust) */
  pwm[0] = -tq[0] - tq[1] + tq[2] + th;
  pwm[1] = +tq[0] + tq[1] + tq[2] + th;
  pwm[2] = -tq[0] + tq[1] - tq[2] + th;
  pwm[3] = +tq[0] - tq[1] - tq[2] + th;
  /* saturation handling */
  real hi = fmax(fmax(pwm[0], pwm[1]), fmax(pwm[2], pwm[3]));
  if (hi > (real)1) for (int i = 0; i < 4; ++i) pwm[i] /= hi;
  real lo = fmin(fmin(pwm[0], pwm[1]), fmin(pwm[2], pwm[3]));
  real fl = (real)q->pwm_floor;
  if (lo < fl) for (int i = 0; i < 4; ++i) pwm[i] += ((real)1 - pwm[i]) / ((real)1 - lo) * (fl - lo);
}

/* Motors.physics_update + BoringBodies.physics_update + rotational drag: body-frame force and
 * torque for this sub-step.  noise[4] are standard normals (zeros when motor_noise is off). */
static void actuate(const te_config* c, ote_drone* d, const real pwm[4], const real noise[4], real Fb[3], real Tb[3]) {
  const te_quad_params* q = &c->quad;
  const real k = (real)c->physics_dt / (real)q->motor_tau;
  /* propeller layout consistent with the mix: m0 front-right, m1 back-left, m2 back-right, m3 front-left */
  const real px[4] = {+1, -1, -1, +1}, py[4] = {-1, +1, -1, +1};
  /* reaction-torque sign: a positive yaw command (motors 0,1 up) must yield +z torque */
  const real ts[4] = {+1, +1, -1, -1};
  const real max_rpm2 = (real)q->total_thrust / ((real)4 * (real)q->thrust_coef);
  real fz = 0, tx = 0, ty = 0, tz = 0;
  for (int i = 0; i < 4; ++i) {
    real t = d->throttle[i];
    t += k * (pwm[i] - t);
    t += noise[i] * t * (real)q->noise_ratio;
    d->throttle[i] = t;
    real rpm2 = t * t * max_rpm2;
    real thrust = (real)q->thrust_coef * rpm2;
    real torque = (real)q->torque_coef * rpm2 * ts[i];
    fz += thrust;
    tx += py[i] * (real)q->arm * thrust;  /* r x F, F = (0,0,thrust) */
    ty += -px[i] * (real)q->arm * thrust;
    tz += torque;
  }
  /* body drag, per body axis: -1/2 rho A Cd |v| v ; rotational: -k |w| w */
  real kd = (real)0.5 * (real)q->air_density * (real)q->drag_area_xyz * (real)q->drag_coef_xyz;
  Fb[0] = -kd * fabs(d->obs_vel[0]) * d->obs_vel[0];
  Fb[1] = -kd * fabs(d->obs_vel[1]) * d->obs_vel[1];
  Fb[2] = -kd * fabs(d->obs_vel[2]) * d->obs_vel[2] + fz;
  real kr = (real)q->drag_coef_pqr;
  Tb[0] = tx - kr * fabs(d->obs_rate[0]) * d->obs_rate[0];
  Tb[1] = ty - kr * fabs(d->obs_rate[1]) * d->obs_rate[1];
  Tb[2] = tz - kr * fabs(d->obs_rate[2]) * d->obs_rate[2];
}

/* BulletClient.stepSimulation for one free body (level4_simulation.py:96): semi-implicit Euler,
 * gyroscopic term, exponential-map attitude update, no damping (PyFlyt disables it). */
static void integrate(const te_config* c, ote_drone* d, const real Fw[3], const real Tw[3]) {
  const te_quad_params* q = &c->quad;
  const real dt = (real)c->physics_dt;
  real m[9];
  quat_to_mat(d->quat, m);
  /* angular: body components  w' = w + dt I^-1 (tau - w x I w) */
  real wb[3], tb[3];
  matT_vec(m, d->omega, wb);
  matT_vec(m, Tw, tb);
  real Iw[3] = {(real)q->inertia[0] * wb[0], (real)q->inertia[1] * wb[1], (real)q->inertia[2] * wb[2]};
  real gy[3] = {wb[1] * Iw[2] - wb[2] * Iw[1], wb[2] * Iw[0] - wb[0] * Iw[2], wb[0] * Iw[1] - wb[1] * Iw[0]};
  real wn[3];
  for (int i = 0; i < 3; ++i) wn[i] = wb[i] + dt * (tb[i] - gy[i]) / (real)q->inertia[i];
  mat_vec(m, wn, d->omega);
  /* linear */
  real inv_m = (real)1 / (real)q->mass;
  d->vel[0] += dt * Fw[0] * inv_m;
  d->vel[1] += dt * Fw[1] * inv_m;
  d->vel[2] += dt * (Fw[2] * inv_m - (real)q->gravity);
  for (int i = 0; i < 3; ++i) d->pos[i] += dt * d->vel[i];
  if (c->ground_contact) { /* opt-in ground plane at ground_z (plane.urdf, entities_manager.py:120-124): inelastic normal contact,
                              Coulomb friction 0.5 against the normal impulse, no contact torque.  Parity unpinned (needs Bullet). */
    const real rest = (real)c->ground_z + (real)c->hull_half_height;
    if (d->pos[2] < rest) {
      real jn = d->vel[2] < 0 ? -d->vel[2] : 0;
      d->pos[2] = rest;
      if (d->vel[2] < 0) d->vel[2] = 0;
      real vt = sqrt(d->vel[0] * d->vel[0] + d->vel[1] * d->vel[1]);
      if (vt > 0) {
        real keep = vt - (real)0.5 * jn;
        keep = keep > 0 ? keep / vt : 0;
        d->vel[0] *= keep; d->vel[1] *= keep;
      }
    }
  }
  /* attitude: q <- exp(dt w / 2) * q, world-frame w */
  real wmag = norm3(d->omega);
  real dq[4];
  real half = (real)0.5 * wmag * dt;
  real sc; /* sin(half)/wmag */
  if (wmag < (real)1e-6) sc = (real)0.5 * dt * ((real)1 - half * half / (real)6);
  else sc = sin(half) / wmag;
  dq[0] = d->omega[0] * sc; dq[1] = d->omega[1] * sc; dq[2] = d->omega[2] * sc; dq[3] = cos(half);
  real x = dq[3] * d->quat[0] + dq[0] * d->quat[3] + dq[1] * d->quat[2] - dq[2] * d->quat[1];
  real y = dq[3] * d->quat[1] - dq[0] * d->quat[2] + dq[1] * d->quat[3] + dq[2] * d->quat[0];
  real z = dq[3] * d->quat[2] + dq[0] * d->quat[1] - dq[1] * d->quat[0] + dq[2] * d->quat[3];
  real w = dq[3] * d->quat[3] - dq[0] * d->quat[0] - dq[1] * d->quat[1] - dq[2] * d->quat[2];
  real n = sqrt(x * x + y * y + z * z + w * w);
  d->quat[0] = x / n; d->quat[1] = y / n; d->quat[2] = z / n; d->quat[3] = w / n;
}

/* Motors.physics_update noise: 4 standard normals per (env, slot, step, sub-step).  One Philox4x32-7 call
 * serves TWO consecutive sub-steps: its 128 bits are split into eight 16-bit uniforms = four Box-Muller
 * pairs (the reference draws np_random.randn(4) from an unseeded stream, quadcopter.py:137,181; any N(0,1)
 * source restates it).  Sub-step `sub` uses words {0,1} when even, {2,3} when odd, of call index sub >> 1. */
static void motor_noise(const ote_env* E, int e, int slot, uint32_t step_index, int sub, real out[4]) {
  if (!E->cfg.motor_noise) { out[0] = out[1] = out[2] = out[3] = 0; return; }
  uint32_t r[4];
  {  /* the counter / key of ote_rng(E, e, OTE_RNG_MOTOR, slot, sub >> 1, episode, step_index), 7 rounds */
    const uint64_t g = (uint64_t)E->cfg.env_index_base + (uint64_t)e;
    const uint32_t ctr[4] = {(uint32_t)g, OTE_RNG_MOTOR | ((uint32_t)slot << 8) | ((uint32_t)(sub >> 1) << 16) | ((uint32_t)(g >> 32) << 24),
                             (uint32_t)E->envs[e].episode, step_index};
    const uint32_t key[2] = {(uint32_t)E->cfg.seed, (uint32_t)(E->cfg.seed >> 32)};
    philox4x32_r(ctr, key, r, 7);
  }
  const uint32_t a = r[(sub & 1) * 2 + 0], b = r[(sub & 1) * 2 + 1];
  const real k16 = (real)(1.0 / 65536.0);
  real r0 = sqrt((real)-2 * log(((real)(a & 0xFFFFu) + (real)0.5) * k16)), a0 = (real)2 * OTE_PI * ((real)(a >> 16) * k16);
  real r1 = sqrt((real)-2 * log(((real)(b & 0xFFFFu) + (real)0.5) * k16)), a1 = (real)2 * OTE_PI * ((real)(b >> 16) * k16);
  out[0] = r0 * cos(a0); out[1] = r0 * sin(a0); out[2] = r1 * cos(a1); out[3] = r1 * sin(a1);
}

/* One physics sub-step of one armed drone: L{2,4}AviarySimulation.step inner body
 * (level4_simulation.py:87-98): update_imu -> update_control -> update_physics -> stepSimulation */
static void substep(const ote_env* E, int e, int slot, ote_drone* d, int mode, uint32_t step_index, int sub) {
  const te_config* c = &E->cfg;
  real nz[4], Fb[3], Tb[3], Fw[3], Tw[3], m[9];
  observe(d);
  /* the reference's loop calls update_control on every physics sub-step (level4_simulation.py:92-94); PyFlyt's own Aviary
   * calls it on every (physics_hz / ctrl_hz)-th one (cfg.control_every_substep == 0) and the motors keep the last pwm */
  int ratio = (int)((real)c->control_dt / (real)c->physics_dt + (real)0.5);
  if (ratio < 1) ratio = 1;
  if (c->control_every_substep || sub % ratio == 0) control(c, d, mode, d->pwm);
  motor_noise(E, e, slot, step_index, sub, nz);
  actuate(c, d, d->pwm, nz, Fb, Tb);
  quat_to_mat(d->quat, m);
  mat_vec(m, Fb, Fw);
  mat_vec(m, Tb, Tw);
  for (int i = 0; i < 3; ++i) { Fw[i] += d->pending[i]; Tw[i] += d->pending[3 + i]; d->pending[i] = 0; d->pending[3 + i] = 0; }
  integrate(c, d, Fw, Tw);
}

/* ------------------------------------------------------------------------- */
/* L2: entity operations (quadcopter.py)                                                        */
/* ------------------------------------------------------------------------- */

/* Quadcopter.convert_command_to_setpoint (quadcopter.py:379-396) */
static void command_to_setpoint(const real cmd[4], real sp[4]) {
  real n = sqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
  real inv = (real)1 / (n > 0 ? n : (real)1);
  sp[0] = cmd[3] * (cmd[0] * inv);
  sp[1] = cmd[3] * (cmd[1] * inv);
  sp[2] = 0;
  sp[3] = cmd[3] * (cmd[2] * inv);
}
OTE_API void ote_command_to_setpoint(const double* cmd, double* sp) {
  real c[4] = {(real)cmd[0], (real)cmd[1], (real)cmd[2], (real)cmd[3]}, s[4];
  command_to_setpoint(c, s);
  for (int i = 0; i < 4; ++i) sp[i] = (double)s[i];
}

/* Quadcopter.disarm (quadcopter.py:461-478): mass 0 (static), velocities zeroed, PyFlyt body/motors
 * reset, setpoint and pwm zeroed.  PID memories and the last IMU read are NOT touched. */
static void disarm(ote_drone* d) {
  for (int i = 0; i < 3; ++i) { d->vel[i] = 0; d->omega[i] = 0; }
  for (int i = 0; i < 4; ++i) { d->throttle[i] = 0; d->setpoint[i] = 0; }
  for (int i = 0; i < 6; ++i) d->pending[i] = 0;
  d->armed = 0;
}
/* Gun.reset (gun.py:118-124) */
static void gun_reset(const te_config* c, ote_drone* d, int max_munition) {
  d->munition = max_munition;
  d->last_fired = -c->cooldown_steps;
}
/* Quadcopter.arm (quadcopter.py:445-459): mass restored, fresh IMU read, gun reset */
static void arm(const te_config* c, ote_drone* d, int max_munition) {
  d->armed = 1;
  observe(d);
  gun_reset(c, d, max_munition);
}
/* Quadcopter.replace (quadcopter.py:433-439): teleport, identity attitude, Bullet zeroes the base
 * velocity; formation_position follows; IMU refreshed only when armed. */
static void replace_drone(ote_drone* d, const real p[3]) {
  for (int i = 0; i < 3; ++i) { d->pos[i] = p[i]; d->formation[i] = p[i]; d->vel[i] = 0; d->omega[i] = 0; }
  d->quat[0] = 0; d->quat[1] = 0; d->quat[2] = 0; d->quat[3] = 1;
  if (d->armed) observe(d);
}

static int max_munition_of(const te_config* c, int slot) {
  if (slot >= c->n_pursuers) return 10; /* Gun default (gun.py:11); invaders never shoot */
  if (c->task == TE_TASK_STAGE02) return slot == 0 ? c->munition : 10; /* stages.py:118; supporter keeps the Gun default */
  return c->munition;
}

/* ---- Gun (core/entities/quadcopters/components/weapons/gun.py) ----------- */
static int gun_has_munition(const ote_drone* d) { return d->munition > 0; }
/* gun.py:56-75 */
static int gun_is_available(const te_config* c, const ote_drone* d, int step) {
  if (!gun_has_munition(d)) return 1;
  return c->cooldown_steps <= step - d->last_fired;
}
/* gun.py:86-99; `u` is the uniform draw standing in for random.random() */
static int gun_shoot(const te_config* c, ote_drone* d, int step, real u) {
  if (!(gun_is_available(c, d, step) && gun_has_munition(d))) return 0;
  d->munition -= 1;
  d->last_fired = step;
  return u < (real)c->hit_prob;
}
/* gun.py:101-113 */
static void gun_state(const te_config* c, const ote_drone* d, int step, int max_munition, real out[3]) {
  real wait = (real)c->cooldown_steps - (real)(step - d->last_fired);
  if (wait < 0) wait = 0;
  out[0] = (real)d->munition / (real)(max_munition > 0 ? max_munition : 1);
  out[1] = wait / (real)c->cooldown_steps;
  out[2] = (real)gun_is_available(c, d, step);
}

/* scripted gun trace for the golden test: events[i] = {step, shoot?}; draws[i] uniform */
OTE_API void ote_gun_trace(int n, const int32_t* steps, const int32_t* shoot, const double* draws, int32_t munition,
                           int32_t cooldown, double hit_prob, int32_t* hit, int32_t* mun_out, double* state_out) {
  te_config c; memset(&c, 0, sizeof c);
  c.cooldown_steps = cooldown; c.hit_prob = (float)hit_prob;
  ote_drone d; memset(&d, 0, sizeof d);
  gun_reset(&c, &d, munition);
  for (int i = 0; i < n; ++i) {
    hit[i] = shoot[i] ? gun_shoot(&c, &d, steps[i], (real)draws[i]) : 0;
    mun_out[i] = d.munition;
    real s[3];
    gun_state(&c, &d, steps[i], munition, s);
    for (int k = 0; k < 3; ++k) state_out[3 * i + k] = (double)s[k];
  }
}

/* ------------------------------------------------------------------------- */
/* geometry_utils.py:6-29 (cone test; only reachable from the non-air-combat navigator)         */
/* ------------------------------------------------------------------------- */
static real degrees_between(const real a[3], const real b[3]) {
  real dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  real c = dot / (norm3(a) * norm3(b));
  return acos(c) * (real)180 / OTE_PI;
}
static int point_inside_cone(const real p[3], const real apex[3], const real base[3], real degrees) {
  real ab[3] = {base[0] - apex[0], base[1] - apex[1], base[2] - apex[2]};
  real ap[3] = {p[0] - apex[0], p[1] - apex[1], p[2] - apex[2]};
  if (norm3(ap) > norm3(ab)) return 0;
  return degrees_between(ap, ab) <= degrees / (real)2;
}
OTE_API int ote_point_inside_cone(const double* p, const double* apex, const double* base, double degrees) {
  real P[3] = {(real)p[0], (real)p[1], (real)p[2]}, A[3] = {(real)apex[0], (real)apex[1], (real)apex[2]};
  real B[3] = {(real)base[0], (real)base[1], (real)base[2]};
  return point_inside_cone(P, A, B, (real)degrees);
}
OTE_API double ote_degrees_between(const double* a, const double* b) {
  real A[3] = {(real)a[0], (real)a[1], (real)a[2]}, B[3] = {(real)b[0], (real)b[1], (real)b[2]};
  return (double)degrees_between(A, B);
}

/* ------------------------------------------------------------------------- */
/* L3: offsets over the snapshot mask (offsets_handler.py)                                      */
/* ------------------------------------------------------------------------- */
static real dist3(const real a[3], const real b[3]) {
  real d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
  return norm3(d);
}
/* identify_closest_pursuer (offsets_handler.py:228-254): closest pursuer in the snapshot */
static int closest_pursuer(const te_config* c, const ote_drone* dr, uint64_t mask, int inv_slot) {
  int best = -1; real bd = 0;
  for (int p = 0; p < c->n_pursuers; ++p) {
    if (!((mask >> p) & 1u)) continue;
    real d = dist3(dr[p].obs_pos, dr[inv_slot].obs_pos);
    if (best < 0 || d < bd) { best = p; bd = d; }
  }
  return best;
}
/* identify_closest_invader (offsets_handler.py:256-281) */
static int closest_invader(const te_config* c, const ote_drone* dr, uint64_t mask, int pur_slot) {
  if (!((mask >> pur_slot) & 1u)) return -1;
  int best = -1; real bd = 0;
  for (int j = c->n_pursuers; j < c->n_pursuers + c->n_invaders; ++j) {
    if (!((mask >> j) & 1u)) continue;
    real d = dist3(dr[pur_slot].obs_pos, dr[j].obs_pos);
    if (best < 0 || d < bd) { best = j; bd = d; }
  }
  return best;
}
/* identify_closest_ally (offsets_handler.py:167-190) */
static int closest_ally(const te_config* c, const ote_drone* dr, uint64_t mask, int pur_slot) {
  if (!((mask >> pur_slot) & 1u)) return -1;
  int count = 0;
  for (int p = 0; p < c->n_pursuers; ++p) count += (mask >> p) & 1u;
  if (count <= 1) return -1;
  int best = -1; real bd = 0;
  for (int p = 0; p < c->n_pursuers; ++p) {
    if (p == pur_slot || !((mask >> p) & 1u)) continue;
    real d = dist3(dr[p].obs_pos, dr[pur_slot].obs_pos);
    if (best < 0 || d < bd) { best = p; bd = d; }
  }
  return best;
}
static uint64_t armed_mask(const ote_drone* dr, int D) {
  uint64_t m = 0;
  for (int i = 0; i < D; ++i) if (dr[i].armed) m |= (uint64_t)1 << i;
  return m;
}

/* ------------------------------------------------------------------------- */
/* navigators                                                                                   */
/* ------------------------------------------------------------------------- */
static void unit_toward(const real from[3], const real to[3], real speed, real cmd[4]) {
  real v[3] = {to[0] - from[0], to[1] - from[1], to[2] - from[2]};
  real n = norm3(v);
  if (n > 0) { v[0] /= n; v[1] /= n; v[2] /= n; }
  cmd[0] = v[0]; cmd[1] = v[1]; cmd[2] = v[2]; cmd[3] = speed;
}

/* KamikazeNavigator._is_building_path_clear: constant False in the air-combat-only navigator
 * (loitering_munition_navigator_air_combat_only.py:83-96); "no pursuer inside the cone from the
 * invader to the building" in the general one (loitering_munition_navigator.py:78-87). */
static int building_path_clear(const te_config* c, const ote_drone* dr, uint64_t mask, int slot, real degrees) {
  if (!c->kamikaze_cone_check) return 0;
  real b[3] = {(real)c->building_position[0], (real)c->building_position[1], (real)c->building_position[2]};
  for (int p = 0; p < c->n_pursuers; ++p)
    if (((mask >> p) & 1u) && point_inside_cone(dr[p].obs_pos, dr[slot].obs_pos, b, degrees)) return 0;
  return 1;
}

/* KamikazeNavigator.update, air-combat-only variant
 * (loitering_munition_navigator_air_combat_only.py:68-78,138-246): check_transition registers the
 * NEXT state, then the CURRENT state executes.  _is_building_path_clear is constant False (:83-96). */
static void kamikaze_update(const te_config* c, ote_drone* dr, uint64_t mask, int slot) {
  ote_drone* d = &dr[slot];
  int state = d->nav_state;
  int pursuers_alive = 0;
  for (int p = 0; p < c->n_pursuers; ++p) pursuers_alive += (mask >> p) & 1u;
  int next = state;
  real cmd[4];
  if (state == TE_NAV_WAIT) {
    if (building_path_clear(c, dr, mask, slot, (real)60)) next = TE_NAV_COLLIDE_BUILDING;
    else if (pursuers_alive > 0) next = TE_NAV_COLLIDE_WINGMAN; /* transition iff >= 1 pursuer */
    cmd[0] = 0; cmd[1] = 0; cmd[2] = 0; cmd[3] = (real)0.4;    /* WaitState.execute :163 */
  } else if (state == TE_NAV_COLLIDE_WINGMAN) {
    if (pursuers_alive == 0) next = TE_NAV_COLLIDE_BUILDING;
    int p = closest_pursuer(c, dr, mask, slot);
    real target[3] = {0, 0, 0};
    if (p >= 0) { target[0] = dr[p].obs_pos[0]; target[1] = dr[p].obs_pos[1]; target[2] = dr[p].obs_pos[2]; }
    unit_toward(d->obs_pos, target, (real)c->invader_speed, cmd);
  } else {
    if (!building_path_clear(c, dr, mask, slot, (real)45)) next = TE_NAV_COLLIDE_WINGMAN;
    real b[3] = {(real)c->building_position[0], (real)c->building_position[1], (real)c->building_position[2]};
    unit_toward(d->obs_pos, b, (real)c->invader_speed, cmd);
  }
  d->nav_state = next;
  command_to_setpoint(cmd, d->setpoint);
}

/* LoyalWingmanBehaviorTree.update (loyalwingman_navigator.py:79-86,238-352): gun available ->
 * ChaseThreat; else (has munition) -> MoveToFormation; out of munition is "available" (gun.py:69-70)
 * so SacrificeAttack == ChaseThreat. */
static void wingman_update(const te_config* c, ote_drone* dr, uint64_t mask, int slot, int step) {
  ote_drone* d = &dr[slot];
  real cmd[4];
  if (gun_is_available(c, d, step)) {
    int j = closest_invader(c, dr, mask, slot);
    real target[3] = {0, 0, 0};
    if (j >= 0) { target[0] = dr[j].obs_pos[0]; target[1] = dr[j].obs_pos[1]; target[2] = dr[j].obs_pos[2]; }
    unit_toward(d->obs_pos, target, (real)c->ally_speed, cmd);
  } else {
    unit_toward(d->obs_pos, d->formation, (real)c->ally_speed, cmd);
  }
  command_to_setpoint(cmd, d->setpoint);
}

/* scripted navigator scenario for the golden/KAT tests: positions[D][3], armed mask, nav states */
OTE_API void ote_kamikaze_scenario(int P, int I, const double* positions, uint32_t mask, const int32_t* nav_in,
                                   double speed, int cone_check, const double* building, int32_t* nav_out,
                                   double* setpoints) {
  te_config c; memset(&c, 0, sizeof c);
  c.n_pursuers = P; c.n_invaders = I; c.invader_speed = (float)speed;
  c.kamikaze_cone_check = cone_check;
  for (int k = 0; k < 3; ++k) c.building_position[k] = (float)building[k];
  ote_drone dr[OTE_MAX_DRONES]; memset(dr, 0, sizeof dr);
  for (int i = 0; i < P + I; ++i) {
    for (int k = 0; k < 3; ++k) dr[i].obs_pos[k] = (real)positions[3 * i + k];
    dr[i].nav_state = nav_in[i];
  }
  for (int j = P; j < P + I; ++j) {
    if (!((mask >> j) & 1u)) continue;
    kamikaze_update(&c, dr, mask, j);
    nav_out[j] = dr[j].nav_state;
    for (int k = 0; k < 4; ++k) setpoints[4 * j + k] = (double)dr[j].setpoint[k];
  }
}
OTE_API void ote_wingman_scenario(int P, int I, const double* positions, const double* formation, uint32_t mask,
                                  int slot, int32_t munition, int32_t last_fired, int32_t step, int32_t cooldown,
                                  double speed, double* setpoint) {
  te_config c; memset(&c, 0, sizeof c);
  c.n_pursuers = P; c.n_invaders = I; c.ally_speed = (float)speed; c.cooldown_steps = cooldown;
  ote_drone dr[OTE_MAX_DRONES]; memset(dr, 0, sizeof dr);
  for (int i = 0; i < P + I; ++i)
    for (int k = 0; k < 3; ++k) dr[i].obs_pos[k] = (real)positions[3 * i + k];
  for (int k = 0; k < 3; ++k) dr[slot].formation[k] = (real)formation[k];
  dr[slot].munition = munition; dr[slot].last_fired = last_fired;
  wingman_update(&c, dr, mask, slot, step);
  for (int k = 0; k < 4; ++k) setpoint[k] = (double)dr[slot].setpoint[k];
}

/* ------------------------------------------------------------------------- */
/* LIDAR own sphere (fused_lidar.py:143-217, lidar_math.py)                                     */
/* ------------------------------------------------------------------------- */
/* LidarMath.cartesian_to_spherical (lidar_math.py:24-34) */
static void cartesian_to_spherical(const real v[3], real out[3]) {
  real r = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (r == 0) { out[0] = 0; out[1] = 0; out[2] = 0; return; }
  out[0] = r;
  out[1] = acos(clampr(v[2] / r, (real)-1, (real)1));
  out[2] = atan2(v[1], v[0]);
}
/* LidarMath.spherical_to_cartesian (lidar_math.py:16-22) */
static void spherical_to_cartesian(const real s[3], real out[3]) {
  out[0] = s[0] * sin(s[1]) * cos(s[2]);
  out[1] = s[0] * sin(s[1]) * sin(s[2]);
  out[2] = s[0] * cos(s[1]);
}
/* LidarMath.index_from_radian (lidar_math.py:93-96): truncate then clip */
static int index_from_radian(real radian, real lo, real hi, int n) {
  int idx = (int)((radian - lo) / (hi - lo) * (real)n);
  return idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
}
static int theta_index(real th) { return index_from_radian(th, (real)0, OTE_PI, TE_LIDAR_NTHETA); }
static int phi_index(real ph) { return index_from_radian(ph, -OTE_PI, OTE_PI, TE_LIDAR_NPHI); }

OTE_API void ote_cartesian_to_spherical(const double* v, double* out) {
  real a[3] = {(real)v[0], (real)v[1], (real)v[2]}, o[3];
  cartesian_to_spherical(a, o);
  for (int i = 0; i < 3; ++i) out[i] = (double)o[i];
}
OTE_API void ote_spherical_to_cartesian(const double* s, double* out) {
  real a[3] = {(real)s[0], (real)s[1], (real)s[2]}, o[3];
  spherical_to_cartesian(a, o);
  for (int i = 0; i < 3; ++i) out[i] = (double)o[i];
}
OTE_API int ote_theta_index(double th) { return theta_index((real)th); }
OTE_API int ote_phi_index(double ph) { return phi_index((real)ph); }
OTE_API double ote_normalize_distance(double d, double max_radius) {
  return (double)clampr((real)d / (real)max_radius, (real)0, (real)1);
}

/* LidarMath.add_features (lidar_math.py:262-311): features [r_hat, theta, phi, flag, delta];
 * invert=0: closer wins (strict <); invert=1: farther wins unless the cell is still empty (>= 1). */
OTE_API void ote_add_features(float* sphere /*3x13x26*/, int n, const double* feats /*n x 5*/, int invert) {
  for (int i = 0; i < n; ++i) {
    real fd = (real)feats[5 * i + 0];
    int ti = theta_index((real)feats[5 * i + 1]);
    int pi = phi_index((real)feats[5 * i + 2]);
    int cell = ti * TE_LIDAR_NPHI + pi;
    real cur = (real)sphere[cell];
    int overwrite = invert ? (cur < (real)1 ? fd > cur : 1) : (fd < cur);
    if (overwrite) {
      sphere[cell] = (float)fd;
      sphere[TE_LIDAR_CELLS + cell] = (float)feats[5 * i + 3];
      sphere[2 * TE_LIDAR_CELLS + cell] = (float)feats[5 * i + 4];
    }
  }
}

/* LidarMath.reframe with local_vector = 0 (lidar_math.py:53-83): R(q_own)^-1 (p_other - p_own).
 * PerceptionSnapshot casts position/quaternion to float32 (perception_snapshot.py:91-110). */
static void reframe_origin(const real p_other[3], const real p_own[3], const real q_own[4], real out[3]) {
  real rel[3];
  for (int i = 0; i < 3; ++i) rel[i] = (real)(float)p_other[i] - (real)(float)p_own[i];
  real n2 = q_own[0] * q_own[0] + q_own[1] * q_own[1] + q_own[2] * q_own[2] + q_own[3] * q_own[3];
  real qi[4] = {-q_own[0] / n2, -q_own[1] / n2, -q_own[2] / n2, q_own[3] / n2};
  rotate_vector(qi, rel, out);
}

/* FusedLIDAR.update_data own sphere for drone `own` (fused_lidar.py:143-217): every OTHER currently
 * armed drone at its Delta=1 snapshot (= last IMU read), closer wins, flag = type/5, time = 1/10. */
/* floats of one own-sphere observation row: lidar_channels x 13 x 26 (3 = FusedLIDAR own sphere; 2 = the legacy LIDAR's layout
 * without the time plane, sensors/lidar.py:137-145) */
static int lidar_words(const te_config* c) { return (c->lidar_channels == 2 ? 2 : TE_LIDAR_CHANNELS) * TE_LIDAR_CELLS; }
static void own_sphere3(const te_config* c, const ote_drone* dr, int D, int own, float* sphere);
static void own_sphere(const te_config* c, const ote_drone* dr, int D, int own, float* out) {
  float tmp[TE_OBS_LIDAR_WORDS];
  own_sphere3(c, dr, D, own, tmp);
  memcpy(out, tmp, (size_t)lidar_words(c) * sizeof(float));
}
static void own_sphere3(const te_config* c, const ote_drone* dr, int D, int own, float* sphere) {
  for (int i = 0; i < TE_OBS_LIDAR_WORDS; ++i) sphere[i] = 1.0f;
  real q[4];
  quat_from_euler(dr[own].obs_euler, q);
  for (int k = 0; k < 4; ++k) q[k] = (real)(float)q[k];
  for (int j = 0; j < D; ++j) {
    if (j == own || !dr[j].armed) continue;
    real local[3], sph[3];
    reframe_origin(dr[j].obs_pos, dr[own].obs_pos, q, local);
    cartesian_to_spherical(local, sph);
    real rhat = clampr(sph[0] / (real)c->lidar_radius, (real)0, (real)1);
    int cell = theta_index(sph[1]) * TE_LIDAR_NPHI + phi_index(sph[2]);
    if (rhat < (real)sphere[cell]) {
      int type = j < c->n_pursuers ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION;
      sphere[cell] = (float)rhat;
      sphere[TE_LIDAR_CELLS + cell] = (float)((real)type / (real)5);
      sphere[2 * TE_LIDAR_CELLS + cell] = (float)((real)1 / (real)10);
    }
  }
}

/* own sphere from raw poses (golden fixture io_data0.h5 cross-check) */
OTE_API void ote_own_sphere_from_poses(int D, int P, const double* pos, const double* euler_own, int own,
                                       const uint8_t* armed, double lidar_radius, float* sphere) {
  te_config c; memset(&c, 0, sizeof c);
  c.n_pursuers = P; c.n_invaders = D - P; c.lidar_radius = (float)lidar_radius;
  ote_drone dr[OTE_MAX_DRONES]; memset(dr, 0, sizeof dr);
  for (int i = 0; i < D; ++i) {
    for (int k = 0; k < 3; ++k) dr[i].obs_pos[k] = (real)pos[3 * i + k];
    dr[i].armed = armed[i];
  }
  for (int k = 0; k < 3; ++k) dr[own].obs_euler[k] = (real)euler_own[k];
  own_sphere3(&c, dr, D, own, sphere);
}

/* ------------------------------------------------------------------------- */
/* level5: snapshot ring + FusedLIDAR stacked observation                                        */
/* (fused_lidar.py:73-109,143-262,293-326; lidar_buffer.py:10-157; lidar_math.py:186-345)        */
/* Canonical clock (SEMANTICS.md): the entry of env-step s holds the wingman's IMU pose of step s  */
/* and the kept features of its own sphere built from the poses of step s; at step t the entry of */
/* step t has age 1 (normalized_delta 0.1, as in the own sphere), an entry of age a is step t-a+1.*/
/* ------------------------------------------------------------------------- */
static float u2f(uint32_t w) { float f; memcpy(&f, &w, 4); return f; }
static uint32_t f2u(float f) { uint32_t w; memcpy(&w, &f, 4); return w; }
static uint32_t* ring_entry(const ote_env* E, int e, int p, int step) {
  return E->ring + (((size_t)e * E->cfg.n_pursuers + p) * TE_RING_DEPTH + (size_t)(step % TE_RING_DEPTH)) * E->entry_words;
}
static void ring_clear_env(ote_env* E, int e) {
  if (!E->ring) return;
  memset(E->ring + (size_t)e * E->cfg.n_pursuers * TE_RING_DEPTH * E->entry_words, 0,
         (size_t)E->cfg.n_pursuers * TE_RING_DEPTH * E->entry_words * sizeof(uint32_t));
}
/* SnapshotBuffer.get_snapshot(publisher, delta_step = age) at env-step `step` (lidar_buffer.py:79-103,159-166) in the
 * canonical clock: the snapshot of step `step - age + 1`, or NULL when that wingman has none (not born yet, dead, or
 * already slid out of the 10 slots) */
static const uint32_t* ring_lookup(const ote_env* E, int e, int p, int step, int age) {
  const int s = step - (age - 1);
  if (age < 1 || age > TE_RING_DEPTH - 1 || s < 1) return NULL;
  const uint32_t* ent = ring_entry(E, e, p, s);
  return (int)ent[0] == s ? ent : NULL;
}
OTE_API int ote_ring_lookup(const ote_env* E, int e, int p, int age) {
  if (!E->ring) return -1;
  /* a disarmed wingman has no buffer at all: messageHub.terminate -> close_buffer -> remove_publisher
   * (quadcopter.py:461-478, lidar_buffer.py:226-237,336-341); here it simply is no candidate any more */
  if (!E->drones[(size_t)e * E->D + p].armed) return 0;
  const uint32_t* ent = ring_lookup(E, e, p, E->envs[e].step, age);
  return ent ? (int)ent[0] : 0;
}
static real cell_margin(real theta, real phi) {
  real ft = theta / OTE_PI * (real)TE_LIDAR_NTHETA, fp = (phi + OTE_PI) / ((real)2 * OTE_PI) * (real)TE_LIDAR_NPHI;
  real mt = fabs(ft - floor(ft + (real)0.5)) * OTE_PI / (real)TE_LIDAR_NTHETA;
  real mp = fabs(fp - floor(fp + (real)0.5)) * (real)2 * OTE_PI / (real)TE_LIDAR_NPHI;
  return mt < mp ? mt : mp;
}
/* FusedLIDAR.update_data of wingman `own` (fused_lidar.py:143-217) -> its ring entry of this step: the kept
 * feature of every cell (closer wins, lidar_math.py:262-311) as (r_hat, theta, phi, type, publisher). */
static void ring_push(ote_env* E, int e, int own, int step) {
  const te_config* c = &E->cfg;
  const int D = E->D;
  const ote_drone* dr = &E->drones[(size_t)e * D];
  uint32_t* ent = ring_entry(E, e, own, step);
  memset(ent, 0, (size_t)E->entry_words * sizeof(uint32_t));
  real q[4];
  quat_from_euler(dr[own].obs_euler, q);
  for (int k = 0; k < 4; ++k) q[k] = (real)(float)q[k];
  ent[0] = (uint32_t)step;
  for (int k = 0; k < 3; ++k) ent[2 + k] = f2u((float)dr[own].obs_pos[k]);
  for (int k = 0; k < 4; ++k) ent[5 + k] = f2u((float)q[k]);
  int n = 0, cells[OTE_MAX_DRONES];
  for (int j = 0; j < D; ++j) {
    if (j == own || !dr[j].armed) continue;
    real local[3], sph[3];
    reframe_origin(dr[j].obs_pos, dr[own].obs_pos, q, local);
    cartesian_to_spherical(local, sph);
    real rhat = clampr(sph[0] / (real)c->lidar_radius, (real)0, (real)1);
    real m = cell_margin(sph[1], sph[2]);
    if (m < E->margin_stack[e]) E->margin_stack[e] = m;
    int cell = theta_index(sph[1]) * TE_LIDAR_NPHI + phi_index(sph[2]);
    int type = j < c->n_pursuers ? TE_TYPE_LOYALWINGMAN : TE_TYPE_LOITERINGMUNITION;
    int at = -1;
    for (int k = 0; k < n; ++k) if (cells[k] == cell) at = k;
    if (at < 0) {
      if (!(rhat < (real)1)) continue; /* an empty cell holds 1.0: strict '<' */
      at = n++; cells[at] = cell;
    } else if (!(rhat < (real)u2f(ent[TE_RING_HEADER_WORDS + 4 * at]))) continue;
    uint32_t* f = ent + TE_RING_HEADER_WORDS + 4 * at;
    f[0] = f2u((float)rhat); f[1] = f2u((float)sph[1]); f[2] = f2u((float)sph[2]);
    f[3] = (uint32_t)type | ((uint32_t)j << 8);
  }
  ent[1] = (uint32_t)n;
}
static void sphere_ones(float* sp) { for (int i = 0; i < TE_OBS_LIDAR_WORDS; ++i) sp[i] = 1.0f; }
/* sphere of a ring entry seen from its own publisher (= that wingman's own sphere of that step) */
static void entry_own_sphere(const uint32_t* ent, float* sp) {
  sphere_ones(sp);
  for (int k = 0; k < (int)ent[1]; ++k) {
    const uint32_t* f = ent + TE_RING_HEADER_WORDS + 4 * k;
    int cell = theta_index((real)u2f(f[1])) * TE_LIDAR_NPHI + phi_index((real)u2f(f[2]));
    sp[cell] = u2f(f[0]);
    sp[TE_LIDAR_CELLS + cell] = (float)((real)(f[3] & 0xFFu) / (real)5);
    sp[2 * TE_LIDAR_CELLS + cell] = (float)((real)1 / (real)10);
  }
}
/* LidarMath.neighbor_sphere_from_new_frame (lidar_math.py:327-345) = transform_features (:186-260) +
 * add_features with the inverted criterion (:248-259,314-325): farther wins unless the cell is still empty. */
static void neighbor_sphere(ote_env* E, int e, const uint32_t* nb, const uint32_t* own, int own_slot, int age, float* sp) {
  const te_config* c = &E->cfg;
  sphere_ones(sp);
  real pn[3], qn[4], po[3], qo[4];
  for (int k = 0; k < 3; ++k) { pn[k] = (real)u2f(nb[2 + k]); po[k] = (real)u2f(own[2 + k]); }
  for (int k = 0; k < 4; ++k) { qn[k] = (real)u2f(nb[5 + k]); qo[k] = (real)u2f(own[5 + k]); }
  real n2 = qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3];
  real qi[4] = {-qo[0] / n2, -qo[1] / n2, -qo[2] / n2, qo[3] / n2};
  for (int k = 0; k < (int)nb[1]; ++k) {
    const uint32_t* f = nb + TE_RING_HEADER_WORDS + 4 * k;
    if ((int)(f[3] >> 8) == own_slot) continue; /* synthetic echo of self (lidar_math.py:228-232) */
    real s3[3] = {(real)u2f(f[0]) * (real)c->lidar_radius, (real)u2f(f[1]), (real)u2f(f[2])}, cart[3], glob[3], rel[3], loc[3], sph[3];
    spherical_to_cartesian(s3, cart);
    rotate_vector(qn, cart, glob);
    for (int i = 0; i < 3; ++i) rel[i] = glob[i] + pn[i] - po[i];
    rotate_vector(qi, rel, loc);
    cartesian_to_spherical(loc, sph);
    real rhat = clampr(sph[0] / (real)c->lidar_radius, (real)0, (real)1);
    real m = cell_margin(sph[1], sph[2]);
    if (m < E->margin_stack[e]) E->margin_stack[e] = m;
    int cell = theta_index(sph[1]) * TE_LIDAR_NPHI + phi_index(sph[2]);
    real cur = (real)sp[cell];
    if (cur < (real)1 ? rhat > cur : 1) {
      sp[cell] = (float)rhat;
      sp[TE_LIDAR_CELLS + cell] = (float)((real)(f[3] & 0xFFu) / (real)5);
      sp[2 * TE_LIDAR_CELLS + cell] = (float)((real)age / (real)TE_RING_DEPTH);
    }
  }
}
/* the draws of FusedLIDAR.bootstrap / randomize_stack (fused_lidar.py:73-81,246-262; lidar_buffer.py:110-157):
 * n ~ U{1..4}; n distinct armed wingmen (the agent included) without replacement; an age ~ U{1..9} each; a
 * uniform permutation of the six (sphere, valid) pairs.  16 Philox words keyed (STACK, slot 0, sub 0..3,
 * episode, step); an integer in [0, k) is (word * k) >> 32. */
typedef struct { int n, who[4], age[4], perm[TE_STACK_SPHERES]; } stack_draws;
static void draw_stack(const ote_env* E, int e, int observer, int episode, int step, uint32_t armed_pursuers, stack_draws* d) {
  uint32_t w[16];
  for (int k = 0; k < 4; ++k) ote_rng(E, e, OTE_RNG_STACK, (uint32_t)observer, (uint32_t)k, (uint32_t)episode, (uint32_t)step, w + 4 * k);
  int cand[OTE_MAX_DRONES], nc = 0;
  for (int p = 0; p < E->cfg.n_pursuers; ++p) if ((armed_pursuers >> p) & 1u) cand[nc++] = p;
  int want = 1 + (int)(((uint64_t)w[0] * 4u) >> 32);
  d->n = want < nc ? want : nc;
  for (int i = 0; i < d->n; ++i) { /* random.sample: partial Fisher-Yates over the candidates in slot order */
    int j = i + (int)(((uint64_t)w[1 + i] * (uint32_t)(nc - i)) >> 32);
    int t = cand[i]; cand[i] = cand[j]; cand[j] = t;
    d->who[i] = cand[i];
    d->age[i] = 1 + (int)(((uint64_t)w[5 + i] * (uint32_t)(TE_RING_DEPTH - 1)) >> 32);
  }
  for (int i = 0; i < TE_STACK_SPHERES; ++i) d->perm[i] = i;
  for (int i = TE_STACK_SPHERES - 1; i >= 1; --i) { /* random.shuffle */
    int j = (int)(((uint64_t)w[9 + (TE_STACK_SPHERES - 1 - i)] * (uint32_t)(i + 1)) >> 32);
    int t = d->perm[i]; d->perm[i] = d->perm[j]; d->perm[j] = t;
  }
}
OTE_API void ote_stack_draws(const te_config* cfg, int env_local, int episode, int step, uint32_t armed_pursuers, int32_t* out /*15*/) {
  ote_env E; memset(&E, 0, sizeof E); E.cfg = *cfg;
  stack_draws d; memset(&d, 0, sizeof d);
  draw_stack(&E, env_local, 0, episode, step, armed_pursuers, &d);
  out[0] = d.n;
  for (int i = 0; i < 4; ++i) { out[1 + i] = i < d.n ? d.who[i] : -1; out[5 + i] = i < d.n ? d.age[i] : 0; }
  for (int i = 0; i < TE_STACK_SPHERES; ++i) out[9 + i] = d.perm[i];
}
/* FusedLIDAR.read_data of wingman `ob` (the agent = slot 0): [own, neighbours...] -> pad -> shuffle.  out[i] = stack[perm[i]].
 * The draws of observer `ob` are keyed with its slot (Level5DumbMultiObs reads every wingman's LIDAR, level5_dumb_multiobs.py:116-150). */
static void stacked_observation(ote_env* E, int e, int ob, int step, uint64_t armed_now, float* out, uint8_t* mask) {
  const te_config* c = &E->cfg;
  const ote_envrec* er = &E->envs[e];
  float stack[TE_STACK_SPHERES][TE_OBS_LIDAR_WORDS];
  uint8_t valid[TE_STACK_SPHERES];
  int nv = 0;
  memset(valid, 0, sizeof valid);
  const uint32_t* own = step >= 1 ? ring_entry(E, e, ob, step) : NULL;
  if (own && (int)own[0] == step) { /* _build_valid_spheres: nothing at all without an own snapshot (:91-96) */
    entry_own_sphere(own, stack[nv]); valid[nv++] = 1;
    stack_draws d;
    draw_stack(E, e, ob, er->episode, step, (uint32_t)(armed_now & (((uint64_t)1 << c->n_pursuers) - 1u)), &d);
    for (int i = 0; i < d.n; ++i) {
      const uint32_t* nb = ring_lookup(E, e, d.who[i], step, d.age[i]);
      if (!nb) continue; /* get_snapshot -> None (lidar_buffer.py:152-154) */
      neighbor_sphere(E, e, nb, own, ob, d.age[i], stack[nv]); valid[nv++] = 1;
    }
    for (int i = nv; i < TE_STACK_SPHERES; ++i) sphere_ones(stack[i]);
    for (int i = 0; i < TE_STACK_SPHERES; ++i) {
      memcpy(out + (size_t)i * TE_OBS_LIDAR_WORDS, stack[d.perm[i]], sizeof stack[0]);
      mask[i] = valid[d.perm[i]];
    }
  } else {
    for (int i = 0; i < TE_STACK_SPHERES; ++i) { sphere_ones(out + (size_t)i * TE_OBS_LIDAR_WORDS); mask[i] = 0; }
  }
}

/* normalize_inertial_data (level4/components/utils/normalization.py:6-30,61-110) + gun state +
 * last action (exp03_vFinal_environment.py:200-228) */
static void inertial_obs(const te_config* c, const ote_drone* d, int step, int max_mun, float out[TE_OBS_INERTIAL_WORDS]) {
  const real two_pi = (real)2 * OTE_PI;
  for (int i = 0; i < 3; ++i) {
    out[0 + i] = (float)clampr(d->obs_pos[i] / (real)c->dome_radius, (real)-1, (real)1);
    out[3 + i] = (float)clampr(d->obs_vel[i] / (real)c->max_speed, (real)-1, (real)1);
    out[6 + i] = (float)clampr(d->obs_euler[i] / OTE_PI, (real)-1, (real)1);
    out[9 + i] = (float)clampr(d->obs_rate[i] / two_pi, (real)-1, (real)1);
  }
  real g[3];
  gun_state(c, d, step, max_mun, g);
  out[12] = (float)g[0]; out[13] = (float)g[1]; out[14] = (float)g[2];
}
OTE_API void ote_normalize_inertial(const double* pos, const double* vel, const double* att, const double* rate,
                                    double max_speed, double dome_radius, float* out12) {
  te_config c; memset(&c, 0, sizeof c);
  c.max_speed = (float)max_speed; c.dome_radius = (float)dome_radius; c.cooldown_steps = 60;
  ote_drone d; memset(&d, 0, sizeof d);
  for (int i = 0; i < 3; ++i) { d.obs_pos[i] = (real)pos[i]; d.obs_vel[i] = (real)vel[i]; d.obs_euler[i] = (real)att[i]; d.obs_rate[i] = (real)rate[i]; }
  float o[TE_OBS_INERTIAL_WORDS];
  inertial_obs(&c, &d, 0, 1, o);
  memcpy(out12, o, 12 * sizeof(float));
}

/* ------------------------------------------------------------------------- */
/* spawn samplers                                                                               */
/* ------------------------------------------------------------------------- */
/* Task.generate_positions (exp03_vFinal_task.py:584-608): theta ~ U(0, pi);
 * phi ~ U(acos(min(min_z, r)/r), pi/2) if r >= min_z else U(0, pi/2); spherical -> cartesian */
static void level4_position(const te_config* c, real r, real u_theta, real u_phi, real out[3]) {
  real min_z = (real)c->born_min_z;
  real theta = u_theta * OTE_PI;
  real lower = min_z < r ? min_z : r;
  real min_phi = acos(lower / r);
  real phi = (r >= min_z) ? min_phi + u_phi * (OTE_PI / (real)2 - min_phi) : u_phi * (OTE_PI / (real)2);
  out[0] = r * sin(phi) * cos(theta);
  out[1] = r * sin(phi) * sin(theta);
  out[2] = r * cos(phi);
}
/* L3Stage1.generate_positions (level3/components/stages.py:360-376): radius ~ U(r, r_max),
 * theta ~ U(0, 2 pi), phi ~ U(0, pi/2) */
static void stage02_position(real r, real r_max, real u_r, real u_theta, real u_phi, real out[3]) {
  if (r > r_max) r_max = r;
  real radius = r + u_r * (r_max - r);
  real theta = u_theta * (real)2 * OTE_PI;
  real phi = u_phi * OTE_PI / (real)2;
  out[0] = radius * sin(phi) * cos(theta);
  out[1] = radius * sin(phi) * sin(theta);
  out[2] = radius * cos(phi);
}
OTE_API void ote_level4_position(double r, double min_z, double u_theta, double u_phi, double* out) {
  te_config c; memset(&c, 0, sizeof c); c.born_min_z = (float)min_z;
  real o[3];
  level4_position(&c, (real)r, (real)u_theta, (real)u_phi, o);
  for (int i = 0; i < 3; ++i) out[i] = (double)o[i];
}

/* ------------------------------------------------------------------------- */
/* level4 family (exp02 / exp03 / exp04 vFinal)                                                 */
/* ------------------------------------------------------------------------- */
/* Task.setup_round (exp03_vFinal_task.py:180-196): disarm all invaders; teleport + arm the first
 * `round` of them on the born-radius cap. */
/* invaders armed in round r: `round` of them in the exp tasks; min((r - 1) * per_round + initial, max) in Level5DumbMultiObjectTask
 * (level5_dumb_multiobject_task.py:173-184) */
static int invaders_in_round(const te_config* c, int round) {
  int n = (round - 1) * c->invaders_per_round + c->initial_invaders;
  return n < c->n_invaders ? n : c->n_invaders;
}
static void level4_setup_round(ote_env* E, int e, int round) {
  const te_config* c = &E->cfg;
  ote_drone* dr = &E->drones[(size_t)e * E->D];
  ote_envrec* er = &E->envs[e];
  for (int j = c->n_pursuers; j < E->D; ++j) disarm(&dr[j]);
  for (int i = 0; i < invaders_in_round(c, round); ++i) {
    uint32_t r[4];
    ote_rng(E, e, OTE_RNG_SPAWN_INVADER, (uint32_t)(c->n_pursuers + i), 0, (uint32_t)er->episode, (uint32_t)round, r);
    real p[3];
    level4_position(c, (real)c->born_radius, u01(r[0]), u01(r[1]), p);
    ote_drone* d = &dr[c->n_pursuers + i];
    replace_drone(d, p);
    arm(c, d, max_munition_of(c, c->n_pursuers + i));
  }
}
/* OffsetHandler.on_episode_start + navigators reset (exp03_vFinal_task.py:173-175,265-268) */
static void level4_refresh_snapshot(ote_env* E, int e, int reset) {
  ote_drone* dr = &E->drones[(size_t)e * E->D];
  E->envs[e].snap_mask = armed_mask(dr, E->D);
  /* a pursuer's word is its kill counter under cfg.evaluation (TE_D_KILLS): cleared with the episode, not with the wave */
  for (int i = reset ? 0 : E->cfg.n_pursuers; i < E->D; ++i) dr[i].nav_state = TE_NAV_WAIT;
}
/* Env.reset -> Task.on_reset (exp03_vFinal_environment.py:128-146, exp03_vFinal_task.py:255-274) */
static void level4_reset_env(ote_env* E, int e) {
  ring_clear_env(E, e); /* buffer_step_broadcast with step 0 resets every LIDAR buffer (base_lidar.py:62-66) */
  const te_config* c = &E->cfg;
  ote_drone* dr = &E->drones[(size_t)e * E->D];
  ote_envrec* er = &E->envs[e];
  er->episode += 1;
  /* on_episode_end: init_constants / init_globals / disarm_all */
  er->step = 0; er->max_step = c->max_step; er->round = 1; er->info_wave = 1;
  er->agent_kills = 0; er->allies_kills = 0; er->deads = 0;
  if (c->reward_model != TE_REWARD_L5_C1) er->last_dist = (real)c->dome_radius; /* Level5C1FusionTask's last_distance outlives every reset */
  for (int k = 0; k < 4; ++k) er->last_action[k] = 0;
  for (int p = 0; p < c->n_pursuers; ++p) for (int k = 0; k < 4; ++k) dr[p].ext_action[k] = 0; /* exp05_vFinal_task.py:139 (init_globals) */
  for (int i = 0; i < E->D; ++i) disarm(&dr[i]);
  /* on_episode_start */
  level4_setup_round(E, e, er->round);
  for (int p = 0; p < c->n_pursuers; ++p) arm(c, &dr[p], max_munition_of(c, p));
  for (int p = 0; p < c->n_pursuers; ++p) {
    uint32_t r[4];
    ote_rng(E, e, OTE_RNG_SPAWN_PURSUER, (uint32_t)p, 0, (uint32_t)er->episode, 0, r);
    real pos[3];
    level4_position(c, (real)c->pursuer_spawn_radius, u01(r[0]), u01(r[1]), pos);
    replace_drone(&dr[p], pos);
  }
  level4_refresh_snapshot(E, e, 1);
}

/* is pursuer s flown by the caller (ote_set_wingman_actions)?  exp05's ally, or a pursuer of cfg.evaluation's driver mask */
static int driven_externally(const te_config* c, int s) {
  return (c->ally_policy == TE_ALLY_EXTERNAL && s == 1) || ((((uint32_t)c->evaluation) >> (8 + s)) & 1u);
}
static void note_margin(real* m, real value, real threshold) {
  real d = fabs(value - threshold);
  if (d < *m) *m = d;
}
/* decision that changes state / done: recorded in both margins (m[0] = all, m[1] = state) */
static void note_state_margin(real m[2], real value, real threshold) {
  note_margin(&m[0], value, threshold);
  note_margin(&m[1], value, threshold);
}

/* One env.step of the level4 family (exp03_vFinal_environment.py:150-171). */
/* cfg.drone_contact (OPT-IN, off in every preset; parity with PyBullet unpinned): armed drones collide with each other (collision
 * group / mask 1 / 1 while armed, 0 / 0 when disarmed: quadcopter.py:484-496).  Stated model, NOT Bullet's solver: spheres of
 * cfg.contact_radius, resolved ONCE per env.step on the state the sub-step loop leaves (the HIP kernel flies the drones of an env
 * in different wavefronts, which cannot exchange positions between physics sub-steps), pairs in slot order, one pass: the
 * overlapping pair is moved apart along the line of centres (half the overlap each) and an approaching normal velocity is
 * shared out (equal masses, restitution 0: both keep the mean); no friction, no torque.  The IMU reads of this step are not
 * touched (they were taken before, level4_simulation.py:92-96). */
static void drone_contacts(const te_config* c, ote_drone* dr, int D, real mg[2]) {
  const real two_r = (real)2 * (real)c->contact_radius;
  for (int i = 0; i < D; ++i) {
    if (!dr[i].armed) continue;
    for (int j = i + 1; j < D; ++j) {
      if (!dr[j].armed) continue;
      real n[3] = {dr[j].pos[0] - dr[i].pos[0], dr[j].pos[1] - dr[i].pos[1], dr[j].pos[2] - dr[i].pos[2]};
      real d = norm3(n);
      note_state_margin(mg, d, two_r);
      if (!(d < two_r) || d <= (real)0) continue;
      for (int k = 0; k < 3; ++k) n[k] /= d;
      const real push = (real)0.5 * (two_r - d);
      real vn = 0;
      for (int k = 0; k < 3; ++k) vn += (dr[j].vel[k] - dr[i].vel[k]) * n[k];
      const real dv = vn < 0 ? (real)0.5 * vn : (real)0;
      for (int k = 0; k < 3; ++k) {
        dr[i].pos[k] -= push * n[k]; dr[j].pos[k] += push * n[k];
        dr[i].vel[k] += dv * n[k]; dr[j].vel[k] -= dv * n[k];
      }
    }
  }
}

static void level4_step_env(ote_env* E, int e, const float* action, float* lidar, float* inertial, float* last_action,
                            float* reward, uint8_t* done, int32_t* info, float* t_lidar, float* t_inertial,
                            float* t_last_action) {
  const te_config* c = &E->cfg;
  const int D = E->D, P = c->n_pursuers;
  ote_drone* dr = &E->drones[(size_t)e * D];
  ote_envrec* er = &E->envs[e];
  real mg[2] = {(real)1e30, (real)1e30};

  /* (1) agent command (quadcopter.py:398-413) */
  real cmd[4] = {(real)action[0], (real)action[1], (real)action[2], (real)action[3]};
  for (int k = 0; k < 4; ++k) er->last_action[k] = cmd[k];
  const int all_scripted = c->evaluation || c->agent_scripted; /* Evaluation_Task / Level5DumbMultiObjectTask: pursuer 0 obeys the behaviour tree too */
  if (!all_scripted) command_to_setpoint(cmd, dr[0].setpoint); /* EvaluationEnvironment.step(actions_not_used) (evaluation_environment.py:170-187) */

  /* (2) task.on_step_start (exp03_vFinal_task.py:232-244,276-283) on the CURRENT offsets snapshot */
  for (int j = P; j < D; ++j) if (dr[j].armed) kamikaze_update(c, dr, er->snap_mask, j);
  if (all_scripted) { /* Evaluation_Task.drive_lw (evaluation_task.py:257-275), Level5DumbMultiObjectTask.drive_loyalwingmen (:256-266): every armed pursuer */
    for (int p = 0; p < P; ++p) if (dr[p].armed && !driven_externally(c, p)) wingman_update(c, dr, er->snap_mask, p, er->step);
  } else if (dr[0].armed) {
    /* get_armed_pursuers()[1:] : with the agent armed these are the armed allies */
    for (int p = 1; p < P; ++p) {
      if (!dr[p].armed) continue;
      if (c->ally_policy == TE_ALLY_BT) wingman_update(c, dr, er->snap_mask, p, er->step);
      else if (c->ally_policy == TE_ALLY_FROZEN) { real z[4] = {0, 0, 0, 1}; command_to_setpoint(z, dr[p].setpoint); }
    }
  } else {
    for (int p = 2; p < P; ++p)
      if (dr[p].armed && c->ally_policy == TE_ALLY_BT) wingman_update(c, dr, er->snap_mask, p, er->step);
  }

  /* (3) advance_step: 8 x simulation.step() = `substeps` physics sub-steps for every armed drone */
  uint32_t step_index = (uint32_t)er->step;
  for (int s = 0; s < c->substeps; ++s)
    for (int i = 0; i < D; ++i)
      if (dr[i].armed) substep(E, e, i, &dr[i], 6, step_index, s);
  if (!c->observe_lag)
    for (int i = 0; i < D; ++i) if (dr[i].armed) observe(&dr[i]);
  er->step += 1; /* AGENT_STEP_BROADCAST: guns, task and LIDAR buffers see the new step */
  const int step = er->step;

  /* (4) task.on_step_middle (exp03_vFinal_task.py:285-319) */
  const uint64_t S = armed_mask(dr, D); /* offsets over drones armed NOW, before engagement */
  er->snap_mask = S;
  int agent_shots = 0, ally_shots = 0, exploded = 0, pursuer_suicided = 0, agent_suicided = 0;
  /* process_shoot_range_invaders (:392-413): pursuers in id order, closest in-range invader */
  for (int p = 0; p < P; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = -1; real bd = 0;
    for (int j = P; j < D; ++j) {
      if (!((S >> j) & 1u)) continue;
      real d = dist3(dr[p].obs_pos, dr[j].obs_pos);
      note_state_margin(mg, d, (real)c->shoot_range);
      if (d < (real)c->shoot_range && (tgt < 0 || d < bd)) { tgt = j; bd = d; }
    }
    if (tgt < 0) continue;
    uint32_t r[4];
    ote_rng(E, e, OTE_RNG_HIT, (uint32_t)p, 0, (uint32_t)er->episode, (uint32_t)step, r);
    if (gun_shoot(c, &dr[p], step, u01(r[0]))) { /* entities_manager.shoot_by_ids (:238-248) */
      disarm(&dr[tgt]);
      if (p == 0) agent_shots += 1; else ally_shots += 1;
      if (c->evaluation) dr[p].nav_state += 1; /* lw_kills (evaluation_task.py:498-499), kept in the pursuer's TE_D_KILLS word */
    }
  }
  /* process_explosion_range_invaders (:359-390), same (stale) distance matrix */
  for (int p = 0; p < P; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = -1; real bd = 0;
    for (int j = P; j < D; ++j) {
      if (!((S >> j) & 1u)) continue;
      real d = dist3(dr[p].obs_pos, dr[j].obs_pos);
      note_state_margin(mg, d, (real)c->explosion_range);
      if (d < (real)c->explosion_range && (tgt < 0 || d < bd)) { tgt = j; bd = d; }
    }
    if (tgt < 0) continue;
    disarm(&dr[p]);
    disarm(&dr[tgt]);
    if (dr[p].munition == 0 && p == 0) agent_suicided += 1;
    else if (dr[p].munition == 0) pursuer_suicided += 1;
    else exploded += 1;
  }
  er->agent_kills += agent_shots; er->allies_kills += ally_shots; er->deads += exploded;
  /* process_invaders_in_origin (:656-659; offsets_handler.py:341-348); commented out in Evaluation_Task (evaluation_task.py:397) */
  for (int j = P; j < D && (!c->evaluation || (c->evaluation & TE_EVAL_ORIGIN_RULE)); ++j) { /* ... and kept by Level52BTEvaluationTask (level5_2bt_evaluation_task.py:328) */
    if (!((S >> j) & 1u)) continue;
    real n = norm3(dr[j].obs_pos);
    note_state_margin(mg, n, (real)c->origin_range);
    if (n < (real)c->origin_range) disarm(&dr[j]);
  }
  if (c->drone_contact) drone_contacts(c, dr, D, mg);

  /* compute_reward (:423-515) */
  real score = 0, bonus = 0, penalty = 0;
  if (!c->evaluation) { /* Evaluation_Task.compute_reward is 0 (evaluation_task.py:508-515) */
    const ote_drone* ag = &dr[0];
    real g[3];
    gun_state(c, ag, step, max_munition_of(c, 0), g);
    real dist_origin = norm3(ag->obs_pos);
    int ally = closest_ally(c, dr, S, 0);
    int target = closest_invader(c, dr, S, ally < 0 ? 0 : ally);
    real tp[3] = {0, 0, 0};
    if (target >= 0) { tp[0] = dr[target].obs_pos[0]; tp[1] = dr[target].obs_pos[1]; tp[2] = dr[target].obs_pos[2]; }
    real cur = dist3(ag->obs_pos, tp);
    int ready = (g[2] == (real)1) || (g[0] == (real)0);
    const real MAXR = (real)1000;
    if (c->reward_model == TE_REWARD_L5_DUMB) { /* Level5DumbMultiObjectTask.compute_reward (level5_dumb_multiobject_task.py:452-553) */
      const real SAFE = (real)5;
      if (ready) score = -cur;
      else {
        score = cur;                                        /* keep away while reloading ... */
        note_margin(&mg[0], cur, SAFE);
        if (cur < SAFE) penalty += (SAFE - cur) / SAFE * ((real)0.5 * MAXR);
      }
      note_margin(&mg[0], cur - er->last_dist, (real)0.01);
      if (g[2] == (real)0 && g[0] > (real)0 && (cur - er->last_dist) > (real)0.01) bonus += (real)0.1 * MAXR;
      if (agent_shots > 0) bonus += (real)agent_shots * MAXR;
      if (ally_shots > 0 || pursuer_suicided > 0) bonus += (real)0.5 * (real)(ally_shots + pursuer_suicided) * MAXR;
      if (agent_suicided > 0) penalty += (real)2 * (real)agent_suicided * MAXR;
      if (exploded > 0) penalty += MAXR * (real)exploded;
      note_margin(&mg[0], ag->obs_pos[2], (real)-5);
      if (ag->obs_pos[2] < (real)-5) { real f = (real)-5 - ag->obs_pos[2]; penalty += (f < (real)1 ? f : (real)1) * MAXR; }
      int outside = 0;
      for (int p = 0; p < P; ++p)
        if ((S >> p) & 1u) {
          real n = norm3(dr[p].obs_pos);
          note_state_margin(mg, n, (real)c->dome_radius);
          if (n > (real)c->dome_radius) outside += 1;
        }
      if (outside > 0) penalty += MAXR;
      note_margin(&mg[0], dist_origin, (real)c->born_radius - (real)2);
      if (dist_origin > (real)c->born_radius - (real)2) { real o = dist_origin - ((real)c->born_radius - (real)2); penalty += o < MAXR ? o : MAXR; }
      er->last_dist = cur;
      real total = score + bonus - penalty;
      score = clampr(total, (real)-3 * MAXR, (real)3 * MAXR); bonus = 0; penalty = 0;
    } else if (c->reward_model == TE_REWARD_L5_C1) { /* Level5C1FusionTask.compute_reward (level5_c1_fusion_task.py:448-485) */
      /* the agent's OWN closest invader (:458), not the closest ally's */
      int t1 = -1; real bd = 0;
      for (int j = P; j < D; ++j)
        if ((S >> j) & 1u) { real d = dist3(ag->obs_pos, dr[j].obs_pos); if (t1 < 0 || d < bd) { t1 = j; bd = d; } }
      if (!((S >> 0) & 1u)) t1 = -1;
      real d1 = t1 >= 0 ? bd : norm3(ag->obs_pos);
      if (er->last_dist == (real)0) er->last_dist = d1; /* `self.last_distance = distance if not hasattr(...) else self.last_distance` (:467-468): set once */
      note_margin(&mg[0], d1, er->last_dist);
      real r1 = 0;
      if (d1 < er->last_dist) r1 += (real)c->approach_bonus_gain * norm3(ag->obs_vel);
      if (agent_shots > 0) r1 += (real)agent_shots * MAXR;
      if (agent_suicided > 0) r1 -= (real)2 * (real)agent_suicided * MAXR;
      score = clampr(r1, (real)-3 * MAXR, (real)3 * MAXR); bonus = 0; penalty = 0;
    } else {
    note_margin(&mg[0], er->last_dist - cur, (real)0.01);
    if ((real)0.01 < er->last_dist - cur && ready) bonus += (real)c->approach_bonus_gain * norm3(ag->obs_vel);
    er->last_dist = cur;
    score = ready ? -cur : cur * ((real)2 * g[1] - (real)1);
    if (agent_shots > 0 || agent_suicided > 0) bonus += (real)(agent_shots + agent_suicided) * MAXR;
    if (ally_shots > 0 || pursuer_suicided > 0) bonus += (real)0.5 * (real)(ally_shots + pursuer_suicided) * MAXR;
    else if (exploded > 0) penalty += MAXR * (real)exploded;
    note_margin(&mg[0], ag->obs_pos[2], (real)-5);
    if (ag->obs_pos[2] < (real)-5) penalty += ((real)-5 - ag->obs_pos[2]) / (real)1 * MAXR;
    int outside = 0;
    for (int p = 0; p < P; ++p)
      if ((S >> p) & 1u) {
        real n = norm3(dr[p].obs_pos);
        note_state_margin(mg, n, (real)c->dome_radius);
        if (n > (real)c->dome_radius) outside += 1;
      }
    if (outside > 0) penalty += MAXR;
    note_margin(&mg[0], dist_origin, (real)c->born_radius - (real)2);
    if (dist_origin > (real)c->born_radius - (real)2) penalty += dist_origin - (real)c->born_radius - (real)2; /* literal, SURVEY.md C8 */
    }
  }
  real rew = score + bonus - penalty;
  /* increment_max_step (:150-153) */
  if (agent_shots + ally_shots > 0) er->max_step += c->step_increment;
  /* compute_termination (:517-569) */
  int armed_invaders = 0, armed_pursuers = 0;
  for (int j = P; j < D; ++j) armed_invaders += dr[j].armed;
  for (int p = 0; p < P; ++p) armed_pursuers += dr[p].armed;
  int all_rounds_over = (armed_invaders == 0) && (er->round >= c->n_rounds);
  int term = 0;
  if (c->evaluation) { /* evaluation_task.py:519-551 */
    if (c->max_step > 0 && step > er->max_step) term = 1; /* TIME_IS_LIMITED */
    if (all_rounds_over || armed_pursuers == 0) term = 1;
    for (int i = 0; i < D; ++i)
      if ((S >> i) & 1u) {
        real n = norm3(dr[i].obs_pos);
        note_state_margin(mg, n, (real)c->dome_radius);
        if (n > (real)c->dome_radius) term = 1;
      }
  } else if (step > er->max_step) term = 1;
  else if (all_rounds_over) term = 1;
  else {
    for (int i = 0; i < D; ++i)
      if ((S >> i) & 1u) {
        real n = norm3(dr[i].obs_pos);
        if (i >= P) note_state_margin(mg, n, (real)c->dome_radius);
        if (n > (real)c->dome_radius) term = 1;
      }
    if (armed_pursuers == 0) term = 1;
    if (c->agent_death_terminates && !dr[0].armed) term = 1; /* commented out in level5_dumb_multiobject_task.py:600-606 */
    note_state_margin(mg, dr[0].obs_pos[2], (real)-5.99);
    if (dr[0].obs_pos[2] < (real)-5.99) term = 1;
  }

  /* (5) info, (6) observation (agent = pursuer 0) */
  int32_t inf[4] = {er->agent_kills, er->allies_kills, er->deads, er->round};
  er->info_wave = er->round; /* compute_info runs before on_step_end (evaluation_environment.py:178-186) */
  const int to_terminal = term && c->auto_reset; /* SB3: the terminal observation travels in `infos` */
  float* L = to_terminal && t_lidar ? t_lidar : lidar;
  float* In = to_terminal && t_inertial ? t_inertial : inertial;
  float* La = to_terminal && t_last_action ? t_last_action : last_action;
  if (L) own_sphere(c, dr, D, 0, L);
  if (In) inertial_obs(c, &dr[0], step, max_munition_of(c, 0), In);
  if (La) for (int k = 0; k < 4; ++k) La[k] = (float)er->last_action[k];
  if (c->stacked_obs) { /* level5_envrionment.py:312-351: every wingman's update_lidar, then the agent's stack */
    E->margin_stack[e] = (real)1e30;
    const uint64_t armed_now = armed_mask(dr, D);
    for (int p = 0; p < P; ++p) if (dr[p].armed) ring_push(E, e, p, step);
    if (E->out_stacked) {
      const int aside = to_terminal && E->out_t_stacked && E->out_t_mask;
      float* So = (aside ? E->out_t_stacked : E->out_stacked) + (size_t)e * TE_OBS_STACKED_WORDS;
      uint8_t* Mo = (aside ? E->out_t_mask : E->out_mask) + (size_t)e * TE_STACK_SPHERES;
      stacked_observation(E, e, 0, step, armed_now, So, Mo);
    }
    if (E->st_stacked) { /* Level5DumbMultiObs.compute_info (level5_dumb_multiobs.py:116-150): every pursuer's student observation */
      for (int p = 0; p < P; ++p) {
        const size_t row = (size_t)e * P + p;
        stacked_observation(E, e, p, step, armed_now, E->st_stacked + row * TE_OBS_STACKED_WORDS, E->st_mask + row * TE_STACK_SPHERES);
        inertial_obs(c, &dr[p], step, max_munition_of(c, p), E->st_inertial + row * TE_OBS_INERTIAL_WORDS);
        /* pursuer.last_action = the behaviour tree's command of this step: (unit direction, 0.6) (loyalwingman_navigator.py:301,325,350) */
        const real sp_ = (real)c->ally_speed;
        float* la = E->st_last_action + row * 4;
        la[0] = (float)(dr[p].setpoint[0] / sp_); la[1] = (float)(dr[p].setpoint[1] / sp_); la[2] = (float)(dr[p].setpoint[3] / sp_); la[3] = (float)sp_;
        if (!dr[p].armed) for (int k = 0; k < 4; ++k) la[k] = 0.0f;
        E->st_active[row] = dr[p].armed ? 1 : 0;
      }
    }
  }

  /* (7) task.on_step_end (:321-333) */
  if (!term && !all_rounds_over && armed_invaders == 0 && armed_pursuers > 0) {
    er->round += (er->round < c->n_rounds) ? 1 : c->n_rounds; /* advance_round (:155-175) */
    level4_setup_round(E, e, er->round);
    level4_refresh_snapshot(E, e, 0);
  }

  /* (8) VecEnv auto-reset */
  if (term && c->auto_reset) {
    level4_reset_env(E, e);
    if (c->stacked_obs && E->out_stacked) { /* no snapshot yet: six empty spheres, nothing valid */
      for (int i = 0; i < TE_OBS_STACKED_WORDS; ++i) E->out_stacked[(size_t)e * TE_OBS_STACKED_WORDS + i] = 1.0f;
      for (int i = 0; i < TE_STACK_SPHERES; ++i) E->out_mask[(size_t)e * TE_STACK_SPHERES + i] = 0;
    }
    if (E->st_stacked) /* the students' reset observation: empty spheres, the fresh IMU / gun rows, no action yet */
      for (int p = 0; p < P; ++p) {
        const size_t row = (size_t)e * P + p;
        for (int i = 0; i < TE_OBS_STACKED_WORDS; ++i) E->st_stacked[row * TE_OBS_STACKED_WORDS + i] = 1.0f;
        for (int i = 0; i < TE_STACK_SPHERES; ++i) E->st_mask[row * TE_STACK_SPHERES + i] = 0;
        inertial_obs(c, &dr[p], 0, max_munition_of(c, p), E->st_inertial + row * TE_OBS_INERTIAL_WORDS);
        for (int k = 0; k < 4; ++k) E->st_last_action[row * 4 + k] = 0.0f;
        E->st_active[row] = dr[p].armed ? 1 : 0;
      }
    if (lidar) for (int i = 0; i < lidar_words(c); ++i) lidar[i] = 1.0f;
    if (inertial) inertial_obs(c, &dr[0], 0, max_munition_of(c, 0), inertial);
    if (last_action) for (int k = 0; k < 4; ++k) last_action[k] = 0.0f;
  }
  *reward = (float)rew;
  *done = (uint8_t)term;
  for (int k = 0; k < 4; ++k) info[k] = inf[k];
  E->margin[e] = mg[0];
  E->margin_state[e] = mg[1];
}

/* ------------------------------------------------------------------------- */
/* stage02 (level3/pyflyt_level3_environment_v2.py + components/stages.py)                      */
/* ------------------------------------------------------------------------- */
static real stage02_agent_min_distance(const te_config* c, const ote_drone* dr, uint64_t S) {
  /* np.sum(np.min(distances[0], axis=0)): row of the FIRST pursuer in the snapshot */
  int first = -1;
  for (int p = 0; p < c->n_pursuers; ++p) if ((S >> p) & 1u) { first = p; break; }
  if (first < 0) return 0;
  real best = 0; int any = 0;
  for (int j = c->n_pursuers; j < c->n_pursuers + c->n_invaders; ++j) {
    if (!((S >> j) & 1u)) continue;
    real d = dist3(dr[first].obs_pos, dr[j].obs_pos);
    if (!any || d < best) { best = d; any = 1; }
  }
  return best;
}
static void stage02_respawn_invader(ote_env* E, int e, int slot, uint32_t tag) {
  uint32_t r[4];
  ote_rng(E, e, OTE_RNG_RESPAWN, (uint32_t)slot, 0, (uint32_t)E->envs[e].episode, tag, r);
  real p[3];
  stage02_position((real)2, (real)6, u01(r[0]), u01(r[1]), u01(r[2]), p); /* stages.py:378-384 */
  replace_drone(&E->drones[(size_t)e * E->D + slot], p);
}
/* on_reset (stages.py:104-131) */
static void stage02_reset_env(ote_env* E, int e) {
  const te_config* c = &E->cfg;
  ote_drone* dr = &E->drones[(size_t)e * E->D];
  ote_envrec* er = &E->envs[e];
  er->episode += 1;
  er->step = 0; er->max_step = c->max_step; er->round = 0;
  er->agent_kills = 0; er->allies_kills = 0; er->deads = 0;
  for (int k = 0; k < 4; ++k) er->last_action[k] = 0;
  for (int i = 0; i < E->D; ++i) disarm(&dr[i]);
  for (int j = c->n_pursuers; j < E->D; ++j) stage02_respawn_invader(E, e, j, 0u);
  for (int p = 0; p < c->n_pursuers; ++p) {
    uint32_t r[4];
    ote_rng(E, e, OTE_RNG_SPAWN_PURSUER, (uint32_t)p, 0, (uint32_t)er->episode, 0, r);
    real pos[3];
    stage02_position((real)c->pursuer_spawn_radius, 0, u01(r[0]), u01(r[1]), u01(r[2]), pos);
    replace_drone(&dr[p], pos);
  }
  for (int i = 0; i < E->D; ++i) arm(c, &dr[i], max_munition_of(c, i));
  er->snap_mask = armed_mask(dr, E->D);
  er->prev_snap_min = stage02_agent_min_distance(c, dr, er->snap_mask);
  er->last_dist = er->prev_snap_min;
}
static void stage02_step_env(ote_env* E, int e, const float* action, float* lidar, float* inertial, float* last_action,
                             float* reward, uint8_t* done, int32_t* info, float* t_lidar, float* t_inertial,
                             float* t_last_action) {
  const te_config* c = &E->cfg;
  const int D = E->D, P = c->n_pursuers;
  ote_drone* dr = &E->drones[(size_t)e * D];
  ote_envrec* er = &E->envs[e];
  real mg[2] = {(real)1e30, (real)1e30};
  real cmd[4] = {(real)action[0], (real)action[1], (real)action[2], (real)action[3]};
  for (int k = 0; k < 4; ++k) er->last_action[k] = cmd[k];
  command_to_setpoint(cmd, dr[0].setpoint);
  /* on_step_start: drive_invaders hover command [0,0,0,0.5] when any pursuer is armed
   * (level3/components/quadcopter_manager.py:175-194); the supporter is never driven (:196-205) */
  int armed_pursuers = 0;
  for (int p = 0; p < P; ++p) armed_pursuers += dr[p].armed;
  if (armed_pursuers > 0)
    for (int j = P; j < D; ++j)
      if (dr[j].armed) { real h[4] = {0, 0, 0, (real)c->invader_speed}; command_to_setpoint(h, dr[j].setpoint); }
  uint32_t step_index = (uint32_t)er->step;
  for (int s = 0; s < c->substeps; ++s)
    for (int i = 0; i < D; ++i)
      if (dr[i].armed) substep(E, e, i, &dr[i], 6, step_index, s);
  if (!c->observe_lag)
    for (int i = 0; i < D; ++i) if (dr[i].armed) observe(&dr[i]);
  er->step += 1;
  const int step = er->step;
  /* on_step_middle (stages.py:144-179) */
  const uint64_t S = armed_mask(dr, D);
  er->snap_mask = S;
  int shots = 0, exploded = 0;
  for (int p = 0; p < P; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = -1; real bd = 0;
    for (int j = P; j < D; ++j) {
      if (!((S >> j) & 1u)) continue;
      real d = dist3(dr[p].obs_pos, dr[j].obs_pos);
      note_state_margin(mg, d, (real)c->shoot_range);
      if (d < (real)c->shoot_range && (tgt < 0 || d < bd)) { tgt = j; bd = d; }
    }
    if (tgt < 0) continue;
    /* shoot_by_ids with the suicide rule (level3/components/quadcopter_manager.py:155-171) */
    if (dr[p].munition == 0) { disarm(&dr[tgt]); shots += 1; continue; }
    uint32_t r[4];
    ote_rng(E, e, OTE_RNG_HIT, (uint32_t)p, 0, (uint32_t)er->episode, (uint32_t)step, r);
    if (gun_shoot(c, &dr[p], step, u01(r[0]))) { disarm(&dr[tgt]); shots += 1; }
  }
  for (int p = 0; p < P; ++p) {
    if (!((S >> p) & 1u)) continue;
    int tgt = -1; real bd = 0;
    for (int j = P; j < D; ++j) {
      if (!((S >> j) & 1u)) continue;
      real d = dist3(dr[p].obs_pos, dr[j].obs_pos);
      note_state_margin(mg, d, (real)c->explosion_range);
      if (d < (real)c->explosion_range && (tgt < 0 || d < bd)) { tgt = j; bd = d; }
    }
    if (tgt < 0) continue;
    disarm(&dr[p]); disarm(&dr[tgt]); exploded += 1;
  }
  er->agent_kills += shots; er->deads += exploded;
  /* compute_reward (stages.py:241-300) */
  real g[3];
  gun_state(c, &dr[0], step, max_munition_of(c, 0), g);
  real cur = stage02_agent_min_distance(c, dr, S);
  real last = er->prev_snap_min;
  real score, bonus = 0, penalty = 0;
  if (g[2] == (real)1) score = -cur;
  else if (g[0] == (real)0) score = -cur;
  else score = cur * ((real)2 * g[1] - (real)1);
  note_margin(&mg[0], last - cur, (real)0.01);
  if ((real)0.01 < last - cur && (g[2] == (real)1 || g[0] == (real)0))
    bonus += (real)c->approach_bonus_gain * norm3(dr[0].obs_vel);
  bonus += (real)1000 * (real)shots;
  penalty += (real)1000 * (real)exploded;
  int outside_p = 0, outside_i = 0;
  for (int i = 0; i < D; ++i)
    if ((S >> i) & 1u) {
      real n = norm3(dr[i].obs_pos);
      note_state_margin(mg, n, (real)c->dome_radius);
      if (n > (real)c->dome_radius) { if (i < P) outside_p += 1; else outside_i += 1; }
    }
  if (outside_p > 0) penalty += (real)1000;
  real rew = score + bonus - penalty;
  /* compute_termination (stages.py:302-344) */
  armed_pursuers = 0;
  for (int p = 0; p < P; ++p) armed_pursuers += dr[p].armed;
  int term = (step > er->max_step) || outside_p > 0 || outside_i > 0 || armed_pursuers < P;
  /* observation.  The reference respawns killed invaders inside on_step_middle, i.e. before
   * compute_observation, but a drone armed after the step broadcast only has a Delta=0 snapshot and
   * the sphere reads Delta=1 (lidar_buffer.py:443-447): it is invisible this step.  Building the
   * sphere before the respawn is the same thing. */
  const int to_terminal = term && c->auto_reset; /* SB3: the terminal observation travels in `infos` */
  float* L = to_terminal && t_lidar ? t_lidar : lidar;
  float* In = to_terminal && t_inertial ? t_inertial : inertial;
  float* La = to_terminal && t_last_action ? t_last_action : last_action;
  if (L) own_sphere(c, dr, D, 0, L);
  if (In) inertial_obs(c, &dr[0], step, max_munition_of(c, 0), In);
  if (La) for (int k = 0; k < 4; ++k) La[k] = (float)er->last_action[k];
  /* respawn disarmed invaders (stages.py:167-174) */
  for (int j = P; j < D; ++j)
    if (!dr[j].armed) { stage02_respawn_invader(E, e, j, (uint32_t)step); arm(c, &dr[j], max_munition_of(c, j)); }
  /* on_step_end: last_offsets = current_offsets (the snapshot BEFORE respawn keeps its distances) */
  er->prev_snap_min = cur;
  er->last_dist = cur;
  if (term && c->auto_reset) {
    stage02_reset_env(E, e);
    if (lidar) for (int i = 0; i < lidar_words(c); ++i) lidar[i] = 1.0f;
    if (inertial) inertial_obs(c, &dr[0], 0, max_munition_of(c, 0), inertial);
    if (last_action) for (int k = 0; k < 4; ++k) last_action[k] = 0.0f;
  }
  *reward = (float)rew; *done = (uint8_t)term;
  info[0] = er->agent_kills; info[1] = 0; info[2] = er->deads; info[3] = 0;
  E->margin[e] = mg[0];
  E->margin_state[e] = mg[1];
}

/* ------------------------------------------------------------------------- */
/* stage01 (level2/pyflyt_level2_environment_modified_v2.py)                                    */
/* slots: 0 = RL pursuer, 1 = idle pursuer, 2 = position-hold invader (mode 7)                  */
/* ------------------------------------------------------------------------- */
static void stage01_uniform_cube(ote_env* E, int e, uint32_t purpose, uint32_t slot, uint32_t index, real p[3]) {
  uint32_t r[4];
  ote_rng(E, e, purpose, slot, 0, (uint32_t)E->envs[e].episode, index, r);
  for (int k = 0; k < 3; ++k) p[k] = (real)-1 + (real)2 * u01(r[k]); /* np.random.uniform(-1, 1, 3) */
}
/* QuadcopterManager.replace_invader (level2/components/quadcopter_manager.py:166-179): teleport, raw
 * mode-7 set-point [x, y, 0, z], then ONE extra imu/control/physics update whose wrench stays
 * accumulated in Bullet until the next stepSimulation. */
static void stage01_replace_invader(ote_env* E, int e, const real p[3], uint32_t step_index) {
  const te_config* c = &E->cfg;
  ote_drone* d = &E->drones[(size_t)e * E->D + 2];
  replace_drone(d, p);
  d->setpoint[0] = p[0]; d->setpoint[1] = p[1]; d->setpoint[2] = 0; d->setpoint[3] = p[2];
  observe(d);
  real pwm[4], nz[4], Fb[3], Tb[3], m[9], Fw[3], Tw[3];
  control(c, d, 7, pwm);
  motor_noise(E, e, 2, step_index, 255, nz);
  actuate(c, d, pwm, nz, Fb, Tb);
  quat_to_mat(d->quat, m);
  mat_vec(m, Fb, Fw); mat_vec(m, Tb, Tw);
  for (int k = 0; k < 3; ++k) { d->pending[k] += Fw[k]; d->pending[3 + k] += Tw[k]; }
}
static void stage01_reset_env(ote_env* E, int e) {
  const te_config* c = &E->cfg;
  ote_drone* dr = &E->drones[(size_t)e * E->D];
  ote_envrec* er = &E->envs[e];
  er->episode += 1;
  er->step = 0; er->max_step = c->max_step; er->round = 0;
  er->agent_kills = 0; er->allies_kills = 0; er->deads = 0;
  for (int k = 0; k < 4; ++k) er->last_action[k] = 0;
  for (int i = 0; i < E->D; ++i) if (!dr[i].armed) { dr[i].armed = 1; gun_reset(c, &dr[i], 0); }
  real p[3];
  stage01_uniform_cube(E, e, OTE_RNG_SPAWN_INVADER, 2, 0, p);
  stage01_replace_invader(E, e, p, 0);
  for (int s = 0; s < 2; ++s) {
    stage01_uniform_cube(E, e, OTE_RNG_SPAWN_PURSUER, (uint32_t)s, 0, p);
    replace_drone(&dr[s], p);
    observe(&dr[s]); /* replace_quadcopter refreshes the IMU (:158-164) */
  }
  dr[0].munition = 0; dr[1].munition = 0;
  er->last_dist = dist3(dr[2].obs_pos, dr[0].obs_pos); /* update_last_distance (:219-223) */
  er->snap_mask = armed_mask(dr, E->D);
}
static void stage01_step_env(ote_env* E, int e, const float* action, float* lidar, float* inertial, float* last_action,
                             float* reward, uint8_t* done, int32_t* info, float* t_lidar, float* t_inertial,
                             float* t_last_action) {
  const te_config* c = &E->cfg;
  const int D = E->D;
  ote_drone* dr = &E->drones[(size_t)e * D];
  ote_envrec* er = &E->envs[e];
  real mg[2] = {(real)1e30, (real)1e30};
  er->step += 1; /* step_calls += 1 (:128) */
  real cmd[4] = {(real)action[0], (real)action[1], (real)action[2], (real)action[3]};
  for (int k = 0; k < 4; ++k) er->last_action[k] = cmd[k];
  command_to_setpoint(cmd, dr[0].setpoint);
  uint32_t step_index = (uint32_t)er->step;
  /* simulation.drones order: invader, pursuer0, pursuer1 (independent bodies: order immaterial) */
  for (int s = 0; s < c->substeps; ++s)
    for (int i = 0; i < D; ++i) substep(E, e, i, &dr[i], i == 2 ? 7 : 6, step_index, s);
  if (!c->observe_lag) for (int i = 0; i < D; ++i) observe(&dr[i]);
  /* observation first (:137), then reward / termination */
  float* L; float* In; float* La;
  real d = dist3(dr[2].obs_pos, dr[0].obs_pos);
  real bonus = 0, penalty = 0;
  note_margin(&mg[0], d, er->last_dist);
  if (d < er->last_dist) bonus += (real)c->approach_bonus_gain * norm3(dr[0].obs_vel);
  note_state_margin(mg, d, (real)c->catch_distance);
  if (d < (real)c->catch_distance) bonus += (real)1000;
  note_state_margin(mg, d, (real)c->dome_radius);
  if (d > (real)c->dome_radius) penalty += (real)1000;
  real rew = -d + bonus - penalty;
  real n0 = norm3(dr[0].obs_pos), n2 = norm3(dr[2].obs_pos);
  note_state_margin(mg, n0, (real)c->dome_radius);
  note_state_margin(mg, n2, (real)c->dome_radius);
  int term = (er->step > er->max_step) || n0 > (real)c->dome_radius || n2 > (real)c->dome_radius;
  const int to_terminal = term && c->auto_reset;
  L = to_terminal && t_lidar ? t_lidar : lidar;
  In = to_terminal && t_inertial ? t_inertial : inertial;
  La = to_terminal && t_last_action ? t_last_action : last_action;
  if (L) own_sphere(c, dr, D, 0, L);
  if (In) inertial_obs(c, &dr[0], er->step, 0, In);
  if (La) for (int k = 0; k < 4; ++k) La[k] = (float)er->last_action[k];
  /* replace_invader_if_close (:147-154), update_last_distance */
  if (d < (real)c->catch_distance) {
    real p[3];
    stage01_uniform_cube(E, e, OTE_RNG_RESPAWN, 2, (uint32_t)er->step, p);
    stage01_replace_invader(E, e, p, step_index);
    er->agent_kills += 1;
  }
  er->last_dist = dist3(dr[2].obs_pos, dr[0].obs_pos);
  if (term && c->auto_reset) {
    stage01_reset_env(E, e);
    if (lidar) for (int i = 0; i < lidar_words(c); ++i) lidar[i] = 1.0f;
    if (inertial) inertial_obs(c, &dr[0], 0, 0, inertial);
    if (last_action) for (int k = 0; k < 4; ++k) last_action[k] = 0.0f;
  }
  *reward = (float)rew; *done = (uint8_t)term;
  info[0] = er->agent_kills; info[1] = 0; info[2] = 0; info[3] = 0;
  E->margin[e] = mg[0];
  E->margin_state[e] = mg[1];
}

/* ------------------------------------------------------------------------- */
/* public API                                                                                   */
/* ------------------------------------------------------------------------- */
static void reset_env(ote_env* E, int e) {
  switch (E->cfg.task) {
    case TE_TASK_STAGE01: stage01_reset_env(E, e); break;
    case TE_TASK_STAGE02: stage02_reset_env(E, e); break;
    default: level4_reset_env(E, e); break;
  }
}

OTE_API ote_env* ote_create(const te_config* cfg) {
  if (!cfg || cfg->struct_size != sizeof(te_config)) return NULL;
  int D = cfg->n_pursuers + cfg->n_invaders;
  if (D < 2 || D > OTE_MAX_DRONES || cfg->n_envs < 1) return NULL;
  ote_env* E = (ote_env*)calloc(1, sizeof(ote_env));
  E->cfg = *cfg; E->D = D;
  E->drones = (ote_drone*)calloc((size_t)cfg->n_envs * D, sizeof(ote_drone));
  E->envs = (ote_envrec*)calloc((size_t)cfg->n_envs, sizeof(ote_envrec));
  E->margin = (real*)calloc((size_t)cfg->n_envs, sizeof(real));
  E->margin_state = (real*)calloc((size_t)cfg->n_envs, sizeof(real));
  E->margin_stack = (real*)calloc((size_t)cfg->n_envs, sizeof(real));
  E->entry_words = TE_RING_ENTRY_WORDS(D);
  if (cfg->stacked_obs)
    E->ring = (uint32_t*)calloc((size_t)cfg->n_envs * cfg->n_pursuers * TE_RING_DEPTH * E->entry_words, sizeof(uint32_t));
  for (size_t i = 0; i < (size_t)cfg->n_envs * D; ++i) E->drones[i].quat[3] = 1;
  for (int e = 0; e < cfg->n_envs; ++e) reset_env(E, e);
  return E;
}
OTE_API void ote_destroy(ote_env* E) {
  if (!E) return;
  free(E->drones); free(E->envs); free(E->margin); free(E->margin_state); free(E->margin_stack); free(E->ring); free(E);
}
OTE_API int ote_real_bytes(void) { return (int)sizeof(real); }
OTE_API int ote_reset(ote_env* E, const uint8_t* mask) {
  for (int e = 0; e < E->cfg.n_envs; ++e) if (!mask || mask[e]) reset_env(E, e);
  return 0;
}
OTE_API int ote_observe(ote_env* E, float* lidar, float* inertial, float* last_action) {
  const te_config* c = &E->cfg;
  for (int e = 0; e < c->n_envs; ++e) {
    ote_drone* dr = &E->drones[(size_t)e * E->D];
    ote_envrec* er = &E->envs[e];
    if (lidar) {
      float* L = lidar + (size_t)e * lidar_words(c);
      /* immediately after reset the Delta=1 snapshot does not exist yet: empty sphere */
      if (er->step == 0) for (int i = 0; i < lidar_words(c); ++i) L[i] = 1.0f;
      else own_sphere(c, dr, E->D, 0, L);
    }
    if (inertial) inertial_obs(c, &dr[0], er->step, max_munition_of(c, 0), inertial + (size_t)e * TE_OBS_INERTIAL_WORDS);
    if (last_action) for (int k = 0; k < 4; ++k) last_action[(size_t)e * 4 + k] = (float)er->last_action[k];
  }
  return 0;
}
/* Observation and command of a caller-driven pursuer `w`: Exp05_vFinal_Task.compute_lw_observation / drive_lw_rl_agent
 * (exp05_vFinal_task.py:252-292) for exp05's ally, Evaluation_Task.drive_lw with a `predict` driver (evaluation_task.py:257-310)
 * for the pursuers of the driver mask.  The reference loops over the ARMED pursuers: a dead one is neither observed nor
 * driven. */
OTE_API int ote_observe_wingman(ote_env* E, int w, float* lidar, float* inertial, float* last_action, uint8_t* active) {
  const te_config* c = &E->cfg;
  if (w < 0 || w >= c->n_pursuers || !driven_externally(c, w)) return 1;
  for (int e = 0; e < c->n_envs; ++e) {
    ote_drone* dr = &E->drones[(size_t)e * E->D];
    ote_envrec* er = &E->envs[e];
    if (lidar) {
      float* L = lidar + (size_t)e * lidar_words(c);
      if (er->step == 0) for (int i = 0; i < lidar_words(c); ++i) L[i] = 1.0f; /* as ote_observe */
      else own_sphere(c, dr, E->D, w, L);
    }
    if (inertial) inertial_obs(c, &dr[w], er->step, max_munition_of(c, w), inertial + (size_t)e * TE_OBS_INERTIAL_WORDS);
    if (last_action) for (int k = 0; k < 4; ++k) last_action[(size_t)e * 4 + k] = (float)dr[w].ext_action[k];
    if (active) active[e] = dr[w].armed ? 1 : 0;
  }
  return 0;
}
OTE_API int ote_set_wingman_actions(ote_env* E, int w, const float* actions) {
  const te_config* c = &E->cfg;
  if (w < 0 || w >= c->n_pursuers || !driven_externally(c, w)) return 1;
  for (int e = 0; e < c->n_envs; ++e) {
    ote_drone* dr = &E->drones[(size_t)e * E->D];
    if (!dr[w].armed) continue;
    real cmd[4];
    for (int k = 0; k < 4; ++k) { cmd[k] = (real)actions[(size_t)e * 4 + k]; dr[w].ext_action[k] = cmd[k]; }
    command_to_setpoint(cmd, dr[w].setpoint);
  }
  return 0;
}
OTE_API int ote_observe_ally(ote_env* E, float* lidar, float* inertial, float* last_action, uint8_t* active) {
  if (E->cfg.ally_policy != TE_ALLY_EXTERNAL || E->cfg.n_pursuers != 2) return 1;
  return ote_observe_wingman(E, 1, lidar, inertial, last_action, active);
}
OTE_API int ote_set_ally_actions(ote_env* E, const float* actions) {
  if (E->cfg.ally_policy != TE_ALLY_EXTERNAL || E->cfg.n_pursuers != 2) return 1;
  return ote_set_wingman_actions(E, 1, actions);
}
/* Evaluation_Task.compute_info (evaluation_task.py:553-574): (lw_kills, lw_alive, lw_munitions, current_wave, step) per pursuer */
OTE_API int ote_wingman_info(const ote_env* E, int32_t* out) {
  const te_config* c = &E->cfg;
  if (!c->evaluation) return 1;
  for (int e = 0; e < c->n_envs; ++e)
    for (int p = 0; p < c->n_pursuers; ++p) {
      const ote_drone* d = &E->drones[(size_t)e * E->D + p];
      int32_t* row = out + ((size_t)e * c->n_pursuers + p) * 5;
      row[0] = d->nav_state; row[1] = d->armed ? 1 : 0; row[2] = d->munition; row[3] = E->envs[e].info_wave; row[4] = E->envs[e].step;
    }
  return 0;
}
OTE_API int ote_step(ote_env* E, const float* actions, float* lidar, float* inertial, float* last_action, float* reward,
                     uint8_t* done, int32_t* info, float* t_lidar, float* t_inertial, float* t_last_action, int threads) {
  const int N = E->cfg.n_envs;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
  for (int e = 0; e < N; ++e) {
    const float* a = actions + (size_t)e * 4;
    float* L = lidar ? lidar + (size_t)e * lidar_words(&E->cfg) : NULL;
    float* In = inertial ? inertial + (size_t)e * TE_OBS_INERTIAL_WORDS : NULL;
    float* La = last_action ? last_action + (size_t)e * 4 : NULL;
    float* tL = t_lidar ? t_lidar + (size_t)e * lidar_words(&E->cfg) : NULL;
    float* tI = t_inertial ? t_inertial + (size_t)e * TE_OBS_INERTIAL_WORDS : NULL;
    float* tA = t_last_action ? t_last_action + (size_t)e * 4 : NULL;
    switch (E->cfg.task) {
      case TE_TASK_STAGE01: stage01_step_env(E, e, a, L, In, La, reward + e, done + e, info + 4 * (size_t)e, tL, tI, tA); break;
      case TE_TASK_STAGE02: stage02_step_env(E, e, a, L, In, La, reward + e, done + e, info + 4 * (size_t)e, tL, tI, tA); break;
      default: level4_step_env(E, e, a, L, In, La, reward + e, done + e, info + 4 * (size_t)e, tL, tI, tA); break;
    }
  }
  return 0;
}
/* level5: te_step_stacked / te_observe_stacked */
OTE_API int ote_step_stacked(ote_env* E, const float* actions, float* stacked, uint8_t* mask, float* inertial, float* last_action,
                             float* reward, uint8_t* done, int32_t* info, float* t_stacked, uint8_t* t_mask, float* t_inertial,
                             float* t_last_action, int threads) {
  if (!E->cfg.stacked_obs || !stacked || !mask) return 1;
  E->out_stacked = stacked; E->out_mask = mask; E->out_t_stacked = t_stacked; E->out_t_mask = t_mask;
  int rc = ote_step(E, actions, NULL, inertial, last_action, reward, done, info, NULL, t_inertial, t_last_action, threads);
  E->out_stacked = NULL; E->out_mask = NULL; E->out_t_stacked = NULL; E->out_t_mask = NULL;
  return rc;
}
/* Level5DumbMultiObs: te_step_students */
OTE_API int ote_step_students(ote_env* E, float* stacked, uint8_t* mask, float* inertial, float* last_action, uint8_t* active,
                              float* reward, uint8_t* done, int32_t* info, int threads) {
  if (!E->cfg.stacked_obs || !(E->cfg.agent_scripted || E->cfg.evaluation) || !stacked || !mask || !inertial || !last_action || !active) return 1;
  E->st_stacked = stacked; E->st_mask = mask; E->st_inertial = inertial; E->st_last_action = last_action; E->st_active = active;
  float* zero = (float*)calloc((size_t)E->cfg.n_envs * 4, sizeof(float));
  int rc = ote_step(E, zero, NULL, NULL, NULL, reward, done, info, NULL, NULL, NULL, threads);
  free(zero);
  E->st_stacked = NULL; E->st_mask = NULL; E->st_inertial = NULL; E->st_last_action = NULL; E->st_active = NULL;
  return rc;
}
OTE_API int ote_observe_stacked(ote_env* E, float* stacked, uint8_t* mask, float* inertial, float* last_action) {
  if (!E->cfg.stacked_obs || !stacked || !mask) return 1;
  for (int e = 0; e < E->cfg.n_envs; ++e) {
    const ote_drone* dr = &E->drones[(size_t)e * E->D];
    stacked_observation(E, e, 0, E->envs[e].step, armed_mask(dr, E->D), stacked + (size_t)e * TE_OBS_STACKED_WORDS,
                        mask + (size_t)e * TE_STACK_SPHERES);
  }
  return ote_observe(E, NULL, inertial, last_action);
}
OTE_API int ote_stack_margins(const ote_env* E, double* out) {
  for (int e = 0; e < E->cfg.n_envs; ++e) out[e] = (double)E->margin_stack[e];
  return 0;
}
OTE_API int ote_margins(const ote_env* E, double* out) {
  for (int e = 0; e < E->cfg.n_envs; ++e) out[e] = (double)E->margin[e];
  return 0;
}
OTE_API int ote_state_margins(const ote_env* E, double* out) {
  for (int e = 0; e < E->cfg.n_envs; ++e) out[e] = (double)E->margin_state[e];
  return 0;
}
/* synthetic actions: dir ~ U(-1,1)^3, mag ~ U(0,1) (apps/threatengage_runner/interactive/analyse.py:55-59) */
OTE_API int ote_random_actions(const ote_env* E, float* actions, uint64_t seed, uint64_t step_index) {
  for (int e = 0; e < E->cfg.n_envs; ++e) {
    uint64_t g = (uint64_t)E->cfg.env_index_base + (uint64_t)e;
    uint32_t ctr[4] = {(uint32_t)g, OTE_RNG_ACTION | ((uint32_t)(g >> 32) << 24), (uint32_t)step_index, (uint32_t)(step_index >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    philox4x32_10(ctr, key, r);
    for (int k = 0; k < 3; ++k) actions[4 * (size_t)e + k] = (float)((real)-1 + (real)2 * u01(r[k]));
    actions[4 * (size_t)e + 3] = (float)u01(r[3]);
  }
  return 0;
}

/* state blob <-> records (layout: include/threatengage.h TE_D_* / TE_E_*) */
static void put_f(uint32_t* w, int at, const real* v, int n) {
  for (int i = 0; i < n; ++i) { float f = (float)v[i]; memcpy(&w[at + i], &f, 4); }
}
static void get_f(const uint32_t* w, int at, real* v, int n) {
  for (int i = 0; i < n; ++i) { float f; memcpy(&f, &w[at + i], 4); v[i] = (real)f; }
}
static size_t ring_words(const ote_env* E) {
  return E->ring ? (size_t)E->cfg.n_envs * E->cfg.n_pursuers * TE_RING_DEPTH * E->entry_words : 0;
}
/* drone records, env records, then (level5) the snapshot ring */
OTE_API size_t ote_state_words(const ote_env* E) {
  return (size_t)E->cfg.n_envs * ((size_t)E->D * TE_DRONE_WORDS + TE_ENV_WORDS) + ring_words(E);
}
OTE_API int ote_get_state(const ote_env* E, uint32_t* dst) {
  const size_t ND = (size_t)E->cfg.n_envs * E->D;
  for (size_t i = 0; i < ND; ++i) {
    const ote_drone* d = &E->drones[i];
    uint32_t* w = dst + i * TE_DRONE_WORDS;
    put_f(w, TE_D_POS, d->pos, 3); put_f(w, TE_D_QUAT, d->quat, 4); put_f(w, TE_D_VEL, d->vel, 3);
    put_f(w, TE_D_OMEGA, d->omega, 3); put_f(w, TE_D_THROTTLE, d->throttle, 4);
    put_f(w, TE_D_PID_AV_I, d->av_i, 3); put_f(w, TE_D_PID_AV_E, d->av_e, 3);
    put_f(w, TE_D_PID_LV_I, d->lv_i, 2); put_f(w, TE_D_PID_LV_E, d->lv_e, 2);
    put_f(w, TE_D_PID_ZV_I, &d->zv_i, 1); put_f(w, TE_D_PID_ZV_E, &d->zv_e, 1);
    put_f(w, TE_D_SETPOINT, d->setpoint, 4);
    put_f(w, TE_D_OBS_POS, d->obs_pos, 3); put_f(w, TE_D_OBS_EULER, d->obs_euler, 3);
    put_f(w, TE_D_OBS_VEL, d->obs_vel, 3); put_f(w, TE_D_OBS_RATE, d->obs_rate, 3);
    put_f(w, TE_D_FORMATION, d->formation, 3); put_f(w, TE_D_PENDING, d->pending, 6);
    w[TE_D_ARMED] = (uint32_t)d->armed; w[TE_D_MUNITION] = (uint32_t)d->munition;
    w[TE_D_LAST_FIRED] = (uint32_t)d->last_fired; w[TE_D_NAV_STATE] = (uint32_t)d->nav_state;
  }
  uint32_t* base = dst + ND * TE_DRONE_WORDS;
  for (int e = 0; e < E->cfg.n_envs; ++e) {
    const ote_envrec* r = &E->envs[e];
    uint32_t* w = base + (size_t)e * TE_ENV_WORDS;
    memset(w, 0, TE_ENV_WORDS * 4);
    w[TE_E_STEP] = (uint32_t)r->step; w[TE_E_MAX_STEP] = (uint32_t)r->max_step; w[TE_E_ROUND] = (uint32_t)r->round; w[TE_E_INFO_WAVE] = (uint32_t)r->info_wave;
    put_f(w, TE_E_LAST_DIST, &r->last_dist, 1);
    w[TE_E_AGENT_KILLS] = (uint32_t)r->agent_kills; w[TE_E_ALLIES_KILLS] = (uint32_t)r->allies_kills;
    w[TE_E_DEADS] = (uint32_t)r->deads; w[TE_E_SNAP_MASK] = (uint32_t)r->snap_mask; w[TE_E_SNAP_MASK_HI] = (uint32_t)(r->snap_mask >> 32); w[TE_E_EPISODE] = (uint32_t)r->episode;
    put_f(w, TE_E_LAST_ACTION, r->last_action, 4); put_f(w, TE_E_PREV_SNAP_MIN, &r->prev_snap_min, 1);
  }
  for (int p = 0; p < E->cfg.n_pursuers; ++p)  /* a caller-driven pursuer keeps its driver's last action in its TE_D_ALLY_ACTION words */
    if (driven_externally(&E->cfg, p))
      for (int e = 0; e < E->cfg.n_envs; ++e) put_f(dst + ((size_t)e * E->D + p) * TE_DRONE_WORDS, TE_D_ALLY_ACTION, E->drones[(size_t)e * E->D + p].ext_action, 4);
  if (E->ring) memcpy(base + (size_t)E->cfg.n_envs * TE_ENV_WORDS, E->ring, ring_words(E) * sizeof(uint32_t));
  return 0;
}
OTE_API int ote_set_state(ote_env* E, const uint32_t* src) {
  const size_t ND = (size_t)E->cfg.n_envs * E->D;
  for (size_t i = 0; i < ND; ++i) {
    ote_drone* d = &E->drones[i];
    const uint32_t* w = src + i * TE_DRONE_WORDS;
    get_f(w, TE_D_POS, d->pos, 3); get_f(w, TE_D_QUAT, d->quat, 4); get_f(w, TE_D_VEL, d->vel, 3);
    get_f(w, TE_D_OMEGA, d->omega, 3); get_f(w, TE_D_THROTTLE, d->throttle, 4);
    get_f(w, TE_D_PID_AV_I, d->av_i, 3); get_f(w, TE_D_PID_AV_E, d->av_e, 3);
    get_f(w, TE_D_PID_LV_I, d->lv_i, 2); get_f(w, TE_D_PID_LV_E, d->lv_e, 2);
    get_f(w, TE_D_PID_ZV_I, &d->zv_i, 1); get_f(w, TE_D_PID_ZV_E, &d->zv_e, 1);
    get_f(w, TE_D_SETPOINT, d->setpoint, 4);
    get_f(w, TE_D_OBS_POS, d->obs_pos, 3); get_f(w, TE_D_OBS_EULER, d->obs_euler, 3);
    get_f(w, TE_D_OBS_VEL, d->obs_vel, 3); get_f(w, TE_D_OBS_RATE, d->obs_rate, 3);
    get_f(w, TE_D_FORMATION, d->formation, 3); get_f(w, TE_D_PENDING, d->pending, 6);
    d->armed = (int32_t)w[TE_D_ARMED]; d->munition = (int32_t)w[TE_D_MUNITION];
    d->last_fired = (int32_t)w[TE_D_LAST_FIRED]; d->nav_state = (int32_t)w[TE_D_NAV_STATE];
  }
  const uint32_t* base = src + ND * TE_DRONE_WORDS;
  for (int e = 0; e < E->cfg.n_envs; ++e) {
    ote_envrec* r = &E->envs[e];
    const uint32_t* w = base + (size_t)e * TE_ENV_WORDS;
    r->step = (int32_t)w[TE_E_STEP]; r->max_step = (int32_t)w[TE_E_MAX_STEP]; r->round = (int32_t)w[TE_E_ROUND]; r->info_wave = (int32_t)w[TE_E_INFO_WAVE];
    get_f(w, TE_E_LAST_DIST, &r->last_dist, 1);
    r->agent_kills = (int32_t)w[TE_E_AGENT_KILLS]; r->allies_kills = (int32_t)w[TE_E_ALLIES_KILLS];
    r->deads = (int32_t)w[TE_E_DEADS]; r->snap_mask = (uint64_t)w[TE_E_SNAP_MASK] | ((uint64_t)w[TE_E_SNAP_MASK_HI] << 32); r->episode = (int32_t)w[TE_E_EPISODE];
    get_f(w, TE_E_LAST_ACTION, r->last_action, 4); get_f(w, TE_E_PREV_SNAP_MIN, &r->prev_snap_min, 1);
  }
  for (int p = 0; p < E->cfg.n_pursuers; ++p)
    if (driven_externally(&E->cfg, p))
      for (int e = 0; e < E->cfg.n_envs; ++e) {
        ote_drone* d = &E->drones[(size_t)e * E->D + p];
        get_f(src + ((size_t)e * E->D + p) * TE_DRONE_WORDS, TE_D_ALLY_ACTION, d->ext_action, 4);
        for (int k = 0; k < 6; ++k) d->pending[k] = 0;  /* not a wrench in this layout */
      }
  if (E->ring) memcpy(E->ring, base + (size_t)E->cfg.n_envs * TE_ENV_WORDS, ring_words(E) * sizeof(uint32_t));
  return 0;
}

/* bare physics helper for analytic KATs: n sub-steps of ONE drone with a fixed set-point */
/* the same, from a given z-velocity integrator (a drone that has hovered before: PID memories survive
 * disarm / replace / arm, quadcopter.py:433-478), also returning the IMU read (body velocity, body rates) */
OTE_API int ote_fly_from(const te_config* cfg, int mode, const double* setpoint, int n_substeps, const double* pos0,
                         double zv_i0, double* out_pos, double* out_vel_body, double* out_euler, double* out_rate_body) {
  ote_env E; memset(&E, 0, sizeof E);
  E.cfg = *cfg; E.cfg.motor_noise = 0; E.D = 1;
  ote_drone d; memset(&d, 0, sizeof d);
  d.quat[3] = 1; d.armed = 1; d.zv_i = (real)zv_i0;
  for (int k = 0; k < 3; ++k) d.pos[k] = (real)pos0[k];
  for (int k = 0; k < 4; ++k) d.setpoint[k] = (real)setpoint[k];
  for (int s = 0; s < n_substeps; ++s) {
    substep(&E, 0, 0, &d, mode, 0, s);   /* observe() at its top leaves the IMU read of the state BEFORE this integration */
    for (int k = 0; k < 3; ++k) {
      out_pos[3 * s + k] = (double)d.obs_pos[k]; out_vel_body[3 * s + k] = (double)d.obs_vel[k];
      out_euler[3 * s + k] = (double)d.obs_euler[k]; out_rate_body[3 * s + k] = (double)d.obs_rate[k];
    }
  }
  return 0;
}
/* The general form for tools/physics_fit.py: set-point schedule (set-point k applies from sub-step sp_start[k] on), the hidden
 * controller / motor state at release (hidden[16] = zv_i, zv_e, lv_i[2], lv_e[2], av_i[3], av_e[3], throttle[4]), optional
 * motor noise (noise_env >= 0: the Philox stream of that env index under cfg->seed).  Sub-step s belongs to env-step s / 16. */
OTE_API int ote_fly_hidden(const te_config* cfg, int mode, const double* setpoints, const int32_t* sp_start, int n_sp, int n_substeps,
                           const double* pos0, const double* hidden, int noise_env, double* out_pos, double* out_vel_body,
                           double* out_euler, double* out_rate_body) {
  ote_env E; memset(&E, 0, sizeof E);
  ote_envrec er; memset(&er, 0, sizeof er);
  E.cfg = *cfg; E.cfg.motor_noise = noise_env >= 0; E.D = 1; E.envs = &er;
  E.cfg.env_index_base = noise_env >= 0 ? noise_env : 0;
  ote_drone d; memset(&d, 0, sizeof d);
  d.quat[3] = 1; d.armed = 1;
  d.zv_i = (real)hidden[0]; d.zv_e = (real)hidden[1];
  for (int k = 0; k < 2; ++k) { d.lv_i[k] = (real)hidden[2 + k]; d.lv_e[k] = (real)hidden[4 + k]; }
  for (int k = 0; k < 3; ++k) { d.av_i[k] = (real)hidden[6 + k]; d.av_e[k] = (real)hidden[9 + k]; }
  for (int k = 0; k < 4; ++k) d.throttle[k] = (real)hidden[12 + k];
  for (int k = 0; k < 3; ++k) d.pos[k] = (real)pos0[k];
  int cur = -1;
  const int per_step = cfg->substeps > 0 ? cfg->substeps : 16;
  for (int s = 0; s < n_substeps; ++s) {
    while (cur + 1 < n_sp && sp_start[cur + 1] <= s) { ++cur; for (int k = 0; k < 4; ++k) d.setpoint[k] = (real)setpoints[4 * cur + k]; }
    substep(&E, 0, 0, &d, mode, (uint32_t)(s / per_step), s % per_step);
    for (int k = 0; k < 3; ++k) {
      out_pos[3 * s + k] = (double)d.obs_pos[k]; out_vel_body[3 * s + k] = (double)d.obs_vel[k];
      out_euler[3 * s + k] = (double)d.obs_euler[k]; out_rate_body[3 * s + k] = (double)d.obs_rate[k];
    }
  }
  return 0;
}
OTE_API int ote_fly(const te_config* cfg, int mode, const double* setpoint, int n_substeps, const double* pos0,
                    double* out_pos, double* out_vel, double* out_euler, double* out_throttle) {
  ote_env E; memset(&E, 0, sizeof E);
  E.cfg = *cfg; E.cfg.motor_noise = 0; E.D = 1;
  ote_drone d; memset(&d, 0, sizeof d);
  d.quat[3] = 1; d.armed = 1;
  for (int k = 0; k < 3; ++k) d.pos[k] = (real)pos0[k];
  for (int k = 0; k < 4; ++k) d.setpoint[k] = (real)setpoint[k];
  for (int s = 0; s < n_substeps; ++s) {
    substep(&E, 0, 0, &d, mode, 0, s);
    if (out_pos) for (int k = 0; k < 3; ++k) out_pos[3 * s + k] = (double)d.pos[k];
    if (out_vel) for (int k = 0; k < 3; ++k) out_vel[3 * s + k] = (double)d.vel[k];
    if (out_euler) { real eu[3]; euler_from_quat(d.quat, eu); for (int k = 0; k < 3; ++k) out_euler[3 * s + k] = (double)eu[k]; }
    if (out_throttle) for (int k = 0; k < 4; ++k) out_throttle[4 * s + k] = (double)d.throttle[k];
  }
  return 0;
}
