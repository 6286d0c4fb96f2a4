/*
 * te_config.c — reference constants per task (plain C, no HIP): te_config_default and the
 * algorithmic-bytes formula.  Interface data only; shared by libthreatengage.so and by the
 * test oracle so both are configured from one table.
 *
 * Citations are file:line under the reference's src/ tree.
 */
#include <math.h>
#include <string.h>

#include "../../include/threatengage.h"

#if defined(__GNUC__)
#define TE_API __attribute__((visibility("default")))
#else
#define TE_API
#endif

/* Task.calculate_rounds (level4/components/tasks_management/tasks/exp03_vFinal_task.py:198-226):
 * ceil of the positive root of n^2 + n - 2 * defenders * munition = 0 */
static int calculate_rounds(int defenders, int munition) {
  double total = (double)defenders * (double)munition;
  double root = (-1.0 + sqrt(1.0 + 8.0 * total)) / 2.0;
  return (int)ceil(root);
}

/* Level5_Task.calculate_max_rounds (threatsense/level5/components/tasks_management/tasks/level5_task.py:105-149):
 * ceil of the positive root of R^2 + (2a - 1) R - 2 (pursuers * munition + 1) = 0, a = NUM_INVADERS */
static int calculate_max_rounds(int pursuers, int munition, int initial_invaders) {
  double total = (double)pursuers * (double)munition + 1.0;
  double b = 2.0 * (double)initial_invaders - 1.0;
  return (int)ceil((-b + sqrt(b * b + 8.0 * total)) / 2.0);
}

/* PyFlyt 0.11.1 "cf2x" (models/vehicles/cf2x/cf2x.yaml + cf2x.urdf) as recalled in SURVEY.md Appendix B from the public
 * sources.  UNVERIFIED: pyflyt is absent from this container, and this table FAILS the one PyBullet recording the reference
 * holds (chi^2 / dof 5.3, tools/physics_fit.py): selectable as TE_QUAD_CF2X_RECALLED, no longer the default (round 3). */
static void cf2x_recalled(te_quad_params* q) {
  q->mass = 0.027f;
  q->inertia[0] = 1.4e-5f; q->inertia[1] = 1.4e-5f; q->inertia[2] = 2.17e-5f;
  q->arm = 0.028f;
  q->total_thrust = 0.5886f;
  q->thrust_coef = 3.16e-10f;
  q->torque_coef = 7.94e-12f;
  q->motor_tau = 0.01f;
  q->noise_ratio = 0.02f;
  q->drag_coef_xyz = 1.0f; q->drag_area_xyz = 0.01f; q->drag_coef_pqr = 1.0e-4f;
  q->air_density = 1.225f;
  q->gravity = 9.81f; /* level4_simulation.py:69-70 */
  q->ang_vel_kp[0] = 8.0e-3f; q->ang_vel_kp[1] = 8.0e-3f; q->ang_vel_kp[2] = 1.0e-2f;
  q->ang_vel_ki[0] = 2.5e-7f; q->ang_vel_ki[1] = 2.5e-7f; q->ang_vel_ki[2] = 1.3e-4f;
  q->ang_vel_kd[0] = 1.0e-4f; q->ang_vel_kd[1] = 1.0e-4f; q->ang_vel_kd[2] = 0.0f;
  q->ang_vel_lim[0] = 1.0f; q->ang_vel_lim[1] = 1.0f; q->ang_vel_lim[2] = 1.0f;
  for (int i = 0; i < 3; ++i) { q->ang_pos_kp[i] = 2.0f; q->ang_pos_lim[i] = 3.0f; }
  for (int i = 0; i < 2; ++i) {
    q->lin_vel_kp[i] = 0.8f; q->lin_vel_ki[i] = 0.3f; q->lin_vel_kd[i] = 0.5f; q->lin_vel_lim[i] = 0.4f;
    q->lin_pos_kp[i] = 1.0f; q->lin_pos_lim[i] = 2.0f;
  }
  q->z_pos_kp = 1.0f; q->z_pos_lim = 1.0f;
  q->z_vel_kp = 0.15f; q->z_vel_ki = 1.0f; q->z_vel_kd = 0.015f; q->z_vel_lim = 1.0f;
  q->pwm_floor = 0.05f;
}

/* THE DEFAULT OF EVERY TASK (round 3): the recalled table with the fewest changed entries that reproduce the recorded PyBullet
 * observations (src/core/rl_framework/utils/output/collect_and_save/io_data0.h5) within their motor-noise scatter under the
 * reference's loop as it reads (update_control on every 240 Hz sub-step, level4_simulation.py:92-94): tools/physics_fit.py,
 * DESIGN.md 5, profiles/r03_physics_fit_headline.md.  A fit to 147 numbers with 21 fitted nuisance values, NOT a verified table, and
 * not the only one the recording supports: in units of its OWN (4-5 x smaller) motor-noise scatter this table scores chi^2 / dof 2.8
 * (the recalled one 5.3) and e.g. "120 Hz control + ang_pos_kp x 2 + total_thrust x 2" scores 0.29 — the data do not identify the
 * entries (only the product kp * arm * thrust / inertia of the rate loop and a slower vertical channel).  It is the default because it is
 * the smallest change that passes under the reference's loop as it reads; te_quad_preset(cfg, TE_QUAD_CF2X_RECALLED) restores PyFlyt's
 * table as recalled.  PyBullet parity of the quadrotor dynamics stays UNPINNED under either preset
 * (tests/test_oracle_physics.py asserts both yardsticks). */
static void cf2x_recorded_fit(te_quad_params* q) {
  cf2x_recalled(q);
  /* tools/physics_fit.py: chi^2 / dof 5.3 -> 0.29 against the 147 recorded numbers with these two entries:
   * the roll / pitch rate loop answers ~6x faster than the recalled table makes it (its product kp * arm * thrust / inertia
   * is what the data identifies: the gain is the entry changed here), and the motors follow their command within one
   * physics sub-step (dt / tau = 1.04) */
  q->ang_vel_kp[0] = 4.8e-2f; q->ang_vel_kp[1] = 4.8e-2f;
  q->motor_tau = 0.004f;
}

TE_API int te_quad_preset(te_config* c, int32_t preset) {
  if (!c) return 1;
  if (preset == TE_QUAD_CF2X_RECALLED) cf2x_recalled(&c->quad);
  else if (preset == TE_QUAD_CF2X_RECORDED_FIT) cf2x_recorded_fit(&c->quad);
  else return 2;
  c->quad_preset = preset;
  return 0;
}

TE_API int te_config_default(te_config* c, int32_t task) {
  if (!c) return 1;
  memset(c, 0, sizeof *c);
  c->struct_size = (uint32_t)sizeof *c;
  c->task = task;
  c->control_every_substep = 1; /* level4_simulation.py:92-94: update_control inside the 240 Hz loop */
  c->lidar_channels = TE_LIDAR_CHANNELS; c->io_location = TE_IO_DEVICE;
  c->drone_contact = 0; c->contact_radius = 0.06f; c->quad_preset = TE_QUAD_CF2X_RECORDED_FIT;
  c->initial_invaders = 1; c->invaders_per_round = 1; c->agent_scripted = 0; c->reward_model = TE_REWARD_EXP03; c->agent_death_terminates = 1;
  c->ground_contact = 0; c->ground_z = -6.0f; c->hull_half_height = 0.0125f; /* plane.urdf at z = -6 (entities_manager.py:120-124): opt-in */
  c->n_envs = 1;
  c->seed = 0;
  c->max_speed = (float)(10.0 * 1000.0 / 3600.0); /* quadcopter.py:590-600 */
  c->substeps = 16;            /* int(120/15) sim steps x 240//120 physics updates */
  c->physics_dt = 1.0f / 240.0f;
  c->control_dt = 1.0f / 120.0f;
  c->observe_lag = 1;
  c->shoot_range = 1.0f; c->explosion_range = 0.2f; c->origin_range = 0.2f; c->hit_prob = 0.9f;
  c->cooldown_steps = 60;      /* gun.py:13-25 */
  c->step_increment = 100;
  c->born_radius = 6.0f; c->born_min_z = 4.0f;
  c->invader_speed = 0.4f; c->ally_speed = 0.6f;
  c->building_position[0] = 0.0f; c->building_position[1] = 0.0f; c->building_position[2] = 0.1f;
  c->motor_noise = 1;
  c->auto_reset = 1;
  c->catch_distance = 0.4f;
  cf2x_recorded_fit(&c->quad);
  switch (task) {
    case TE_TASK_STAGE01: /* level2/pyflyt_level2_environment_modified_v2.py:27-73 */
      c->n_pursuers = 2; c->n_invaders = 1;
      c->dome_radius = 10.0f; c->lidar_radius = 20.0f; /* level2/components/quadcopter_manager.py:44 */
      c->munition = 0; c->max_step = 300; c->n_rounds = 0;
      c->pursuer_spawn_radius = 1.0f; c->ally_policy = TE_ALLY_NONE; c->approach_bonus_gain = 10.0f;
      break;
    case TE_TASK_STAGE02: /* level3/components/stages.py:65-83,118 ; env default dome 8 (pyflyt_level3_environment_v2.py:32) */
      c->n_pursuers = 2; c->n_invaders = 5;
      c->dome_radius = 8.0f; c->lidar_radius = 16.0f;
      c->munition = 4; c->max_step = 600; c->n_rounds = 0;
      c->pursuer_spawn_radius = 1.0f; c->ally_policy = TE_ALLY_NONE; c->approach_bonus_gain = 10.0f;
      c->invader_speed = 0.5f; /* hover command magnitude (level3/components/quadcopter_manager.py:191) */
      break;
    case TE_TASK_EXP02: /* exp02_vFinal_task.py:87-111 */
      c->n_pursuers = 1; c->munition = 20;
      c->n_rounds = calculate_rounds(1, 20); c->n_invaders = c->n_rounds;
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 300; c->pursuer_spawn_radius = 2.0f; c->ally_policy = TE_ALLY_NONE; c->approach_bonus_gain = 1.0f;
      break;
    case TE_TASK_EXP03: /* exp03_vFinal_task.py:88-112 */
    case TE_TASK_EXP04: /* exp04_vFinal_task.py (ally frozen, bonus x10) */
    case TE_TASK_EXP05: /* exp05_vFinal_task.py:99-124: the exp03 constants; the ally obeys a second policy (:252-260) */
      c->n_pursuers = 2; c->munition = 20;
      c->n_rounds = calculate_rounds(2, 20); c->n_invaders = c->n_rounds;
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 300; c->pursuer_spawn_radius = 2.0f;
      c->ally_policy = task == TE_TASK_EXP03 ? TE_ALLY_BT : task == TE_TASK_EXP04 ? TE_ALLY_FROZEN : TE_ALLY_EXTERNAL;
      c->approach_bonus_gain = task == TE_TASK_EXP04 ? 10.0f : 1.0f;
      break;
    case TE_TASK_LEVEL5: /* level5_task.py:76-98: the exp03 task logic with 6 wingmen, 12 invader slots, stacked observation */
      c->n_pursuers = 6; c->munition = 20; c->n_invaders = 12;
      c->n_rounds = calculate_max_rounds(6, 20, 12);
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 300; c->pursuer_spawn_radius = 2.0f;
      c->ally_policy = TE_ALLY_BT; c->approach_bonus_gain = 1.0f;
      c->stacked_obs = 1;
      break;
    case TE_TASK_LEVEL5_DUMB: { /* level5_dumb_multiobject_task.py:82-115 */
      const int initial = 5, per_round = 1, max_invaders = 30;
      c->n_pursuers = 6 + 1; c->n_invaders = max_invaders;
      c->initial_invaders = initial; c->invaders_per_round = per_round;
      c->n_rounds = (int)ceil((double)(max_invaders - initial) / per_round + 1.0);       /* 26 */
      c->munition = (initial + max_invaders) * c->n_rounds / 2;                          /* 455 */
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 300; c->pursuer_spawn_radius = 2.0f;
      c->ally_policy = TE_ALLY_BT; c->approach_bonus_gain = 1.0f;
      c->stacked_obs = 1; c->agent_scripted = 1; c->reward_model = TE_REWARD_L5_DUMB; c->agent_death_terminates = 0;
      break;
    }
    case TE_TASK_LEVEL5_FUSION: { /* level5_fusion_task.py:81-112 */
      const int initial = 5, per_round = 5, max_invaders = 30;
      c->n_pursuers = 5 + 1; c->n_invaders = max_invaders;
      c->initial_invaders = initial; c->invaders_per_round = per_round;
      c->n_rounds = (int)ceil((double)(max_invaders - initial) / per_round + 1.0);       /* 6 */
      c->munition = (initial + max_invaders) * c->n_rounds / 2;                          /* 105 */
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 300; c->pursuer_spawn_radius = 2.0f;
      c->ally_policy = TE_ALLY_BT; c->approach_bonus_gain = 1.0f;
      c->stacked_obs = 1; c->reward_model = TE_REWARD_L5_DUMB;
      break;
    }
    case TE_TASK_LEVEL5_C1: { /* level5_c1_fusion_task.py:82-111 */
      const int initial = 4, per_round = 1, max_invaders = 10;
      c->n_pursuers = 1 + 1; c->n_invaders = max_invaders;
      c->initial_invaders = initial; c->invaders_per_round = per_round;
      c->n_rounds = (int)ceil((double)(max_invaders - initial) / per_round + 1.0);       /* 7 */
      c->munition = (initial + max_invaders) * c->n_rounds / 2;                          /* 49 */
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 300; c->pursuer_spawn_radius = 2.0f;
      c->ally_policy = TE_ALLY_BT; c->approach_bonus_gain = 10.0f;                       /* reward += 10 * |v| (:467-468) */
      c->stacked_obs = 1; c->reward_model = TE_REWARD_L5_C1;
      break;
    }
    case TE_TASK_LEVEL5_2BT: { /* level5_2bt_evaluation_task.py:82-111 */
      const int initial = 5, per_round = 1, max_invaders = 30;
      c->n_pursuers = 1 + 1; c->n_invaders = max_invaders;
      c->initial_invaders = initial; c->invaders_per_round = per_round;
      c->n_rounds = (int)ceil((double)(max_invaders - initial) / per_round + 1.0);       /* 26 */
      c->munition = (initial + max_invaders) * c->n_rounds / 2;                          /* 455 */
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 300 + 10 * 100; c->step_increment = 0;                               /* MAX_STEP = 300 + 10 * STEP_INCREMENT, never extended */
      c->pursuer_spawn_radius = 2.0f; c->ally_policy = TE_ALLY_BT; c->approach_bonus_gain = 1.0f;
      c->evaluation = TE_EVAL_ON | TE_EVAL_ORIGIN_RULE;
      break;
    }
    case TE_TASK_EVALUATION: /* evaluation_task.py:89-112: one behaviour-tree driver (evaluation_exp01_1bt_app_ready.py:66-68),
                                TIME_IS_LIMITED False (max_step 0 = no limit), the exp03 constants otherwise */
      c->n_pursuers = 1; c->munition = 20;
      c->n_rounds = calculate_rounds(1, 20); c->n_invaders = c->n_rounds;
      c->dome_radius = 20.0f; c->lidar_radius = 40.0f;
      c->max_step = 0; c->pursuer_spawn_radius = 2.0f;
      c->ally_policy = TE_ALLY_BT; c->approach_bonus_gain = 1.0f;
      c->evaluation = 1;
      break;
    default:
      return 2;
  }
  return 0;
}

/* B(P,I,C) = (P+I)*2*176 + 2*40 + 16 + (C*338*4 + 15*4 + 4*4 + 4 + 4) */
TE_API int te_algorithmic_bytes_per_env_step(const te_config* c, size_t* out) {
  if (!c || !out) return 1;
  size_t D = (size_t)(c->n_pursuers + c->n_invaders);
  *out = D * 2 * 176 + 2 * 40 + 16 + ((size_t)(c->lidar_channels == 2 ? 2 : TE_LIDAR_CHANNELS) * TE_LIDAR_CELLS * 4 + 15 * 4 + 4 * 4 + 4 + 4);
  if (c->stacked_obs) {  /* level5: 6 spheres + mask instead of 1; one ring entry written per wingman, <= 4 read */
    size_t P = (size_t)c->n_pursuers, entry = (size_t)TE_RING_ENTRY_WORDS(D) * 4;
    *out += (size_t)(TE_STACK_SPHERES - 1) * TE_OBS_LIDAR_WORDS * 4 + TE_STACK_SPHERES + P * entry + 4 * entry;
  }
  if (c->ally_policy == TE_ALLY_EXTERNAL)  /* exp05: the ally's observation (sphere + 15 + 4 floats + active byte) written, its action read,
                                              the poses of all D drones read once more (7 words each), 11 words of the ally written */
    *out += (size_t)TE_OBS_LIDAR_WORDS * 4 + 15 * 4 + 4 * 4 + 1 + 4 * 4 + D * 7 * 4 + 11 * 4;
  return 0;
}

TE_API int te_abi_version(void) { return TE_ABI_VERSION; }
TE_API int te_calculate_rounds(int32_t defenders, int32_t munition) { return calculate_rounds(defenders, munition); }
