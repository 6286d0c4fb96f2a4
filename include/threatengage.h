/*
 * threatengage.h — C ABI of the batched threat-engagement drone environment
 * (MI355X / gfx950 HIP implementation: dronechase_amd/libthreatengage.so).
 *
 * The reference (DaviGuanabara/dronechase) has no FFI: every environment is a
 * Python gymnasium.Env driven one-process-per-env through SB3's SubprocVecEnv
 * (src/core/rl_framework/utils/pipeline.py:31-61).  This header CREATES the
 * seam: one `te_env` replaces N reference environments
 * (src/threatengage/environments/level4/exp03_vFinal_environment.py:42-171 and
 * siblings) and each entry point names the reference interface it stands for.
 *
 * Conventions
 *   - plain C, no torch/HIP types in signatures; every I/O pointer is a DEVICE
 *     pointer (hipMalloc'ed or a PyTorch-ROCm tensor's data_ptr()) unless the
 *     name ends in `_host`.
 *   - all functions return 0 on success, non-zero on failure; the message is
 *     available from te_last_error() (thread-local, never NULL).
 *   - work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 *     null stream) and is asynchronous; the caller synchronises.
 *   - a te_env is not thread-safe; distinct te_envs (e.g. one per GPU) are
 *     independent (mirrors the thread-local singletons of the reference,
 *     level4/components/entities_management/entities_manager.py:24-30).
 */
#ifndef THREATENGAGE_H
#define THREATENGAGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TE_ABI_VERSION 5   /* layout of te_config / the state blob.  Added since without a layout change: the task presets 10-12, TE_REWARD_L5_C1,
                              bit 1 of te_config.evaluation (TE_EVAL_ORIGIN_RULE) */

/* ---- tasks (reference env class each one mirrors) ----------------------- */
enum {
  TE_TASK_STAGE01 = 1, /* level2/pyflyt_level2_environment_modified_v2.py (apps stage01) */
  TE_TASK_STAGE02 = 2, /* level3/pyflyt_level3_environment_v2.py + components/stages.py (apps stage02) */
  TE_TASK_EXP02 = 3,   /* level4/exp02_vFinal_environment.py + tasks/exp02_vFinal_task.py */
  TE_TASK_EXP03 = 4,   /* level4/exp03_vFinal_environment.py + tasks/exp03_vFinal_task.py (apps stage03) */
  TE_TASK_EXP04 = 5,   /* level4/exp04_vFinal_environment.py (ally frozen, x10 approach bonus) */
  TE_TASK_LEVEL5 = 6,  /* threatsense/level5/level5_envrionment.py + tasks/level5_task.py: the exp03 task with 6 pursuers,
                          12 invaders and the FusedLIDAR stacked-sphere observation (te_step_stacked) */
  TE_TASK_EXP05 = 7,   /* level4/exp05_vFinal_environment.py + tasks/exp05_vFinal_task.py: exp03 with the ally driven by a
                          second policy (drive_lw_rl_agent, :252-260): te_observe_ally / te_set_ally_actions */
  TE_TASK_LEVEL5_DUMB = 9, /* threatsense/level5/level5_dumb_multiobs.py + tasks/level5_dumb_multiobject_task.py: the imitation-data
                          collector's environment — 7 wingmen ALL flown by the behaviour tree (the agent too, :256-266), 30 invader slots
                          of which min(4 + round, 30) are armed in a round (:173-184), 26 rounds, its own reward (:452-553) and
                          termination (:555-612); the observation is the stacked one of EVERY armed wingman (te_step_students) */
  TE_TASK_LEVEL5_2BT = 10, /* threatsense/level5/level5_eval_2bt_environment.py + tasks/level5_2bt_evaluation_task.py: two wingmen, both flown by
                          the behaviour tree (:256-261), against the 30-slot invader table of the level5 fusion tasks (5 in round 1, one more
                          per round, 26 rounds); Evaluation_Task-style rules (cfg.evaluation = TE_EVAL_ON | TE_EVAL_ORIGIN_RULE): reward 0
                          (:421-428), a fixed limit of 1 300 steps (:111), kills counted per wingman (:127-134,408-409; te_wingman_info);
                          the environment returns no observation (level5_eval_2bt_environment.py:27-33,53-56) */
  TE_TASK_LEVEL5_C1 = 11, /* threatsense/level5/level5_c1_fusion_environment.py + tasks/level5_c1_fusion_task.py: the RL agent and ONE scripted wingman
                          against 10 invader slots (4 in round 1, one more per round, 7 rounds; munition 49), the stacked observation of level5
                          (te_step_stacked) and the task's minimal reward (cfg.reward_model = TE_REWARD_L5_C1) */
  TE_TASK_LEVEL5_FUSION = 12, /* threatsense/level5/level5_fusion_environment.py + tasks/level5_fusion_task.py: the RL agent + 5 scripted wingmen
                          against 30 invader slots, FIVE more per round (5, 10, ... 30: 6 rounds; munition 105), Level5DumbMultiObjectTask's reward
                          (the two compute_reward bodies are the same text) with the agent's death ending the episode; 36 drones per env: the
                          stacked observation through te_step_stacked / te_observe_stacked (te_observe serves at most 32) */
  TE_TASK_EVALUATION = 8 /* level4/evaluation_environment.py + tasks/evaluation_task.py with behaviour-tree drivers only
                          (apps/threatengage_runner/stage03/experiments/01/evaluation_exp01_1bt_app_ready.py): cfg.evaluation = 1,
                          n_pursuers = number of drivers (default 1) */
};

/* ally (pursuer slots >= 1) policy: nobody (the set-point persists), LoyalWingmanBehaviorTree, drive([0,0,0,1]) every
 * step (exp04), or the caller through te_set_ally_actions (exp05; n_pursuers must be 2, exp05_vFinal_task.py:103) */
enum { TE_ALLY_NONE = 0, TE_ALLY_BT = 1, TE_ALLY_FROZEN = 2, TE_ALLY_EXTERNAL = 3 };
enum { TE_EVAL_ON = 1, TE_EVAL_ORIGIN_RULE = 2 };   /* te_config.evaluation, bits 0 and 1 */

/* kamikaze FSM states (core/entities/navigators/loitering_munition_navigator_air_combat_only.py:138-246) */
enum { TE_NAV_WAIT = 0, TE_NAV_COLLIDE_WINGMAN = 1, TE_NAV_COLLIDE_BUILDING = 2 };

/* entity types; LIDAR flag = type / 5 (core/entities/entity_type.py:4-14, lidar_math.py:305) */
enum { TE_TYPE_LOITERINGMUNITION = 1, TE_TYPE_QUADCOPTER = 2, TE_TYPE_LOYALWINGMAN = 3,
       TE_TYPE_PROTECTED_BUILDING = 4, TE_TYPE_GROUND = 5 };

/* ---- LIDAR grid (core/dataclasses/angle_grid.py:28-37, resolution 16) --- */
#define TE_LIDAR_NTHETA 13
#define TE_LIDAR_NPHI 26
#define TE_LIDAR_CELLS (TE_LIDAR_NTHETA * TE_LIDAR_NPHI) /* 338 */
#define TE_LIDAR_CHANNELS 3                              /* distance, flag, time (core/enums/channel_index.py:7-9) */
#define TE_OBS_LIDAR_WORDS (TE_LIDAR_CHANNELS * TE_LIDAR_CELLS) /* 1014 */
#define TE_OBS_INERTIAL_WORDS 15 /* pos3 vel3 att3 rate3 gun3 (exp03_vFinal_environment.py:200-228) */
#define TE_OBS_ACTION_WORDS 4
#define TE_INFO_WORDS 4 /* agent_kills, allies_kills, deads, current_wave (exp03_vFinal_task.py:571-578) */

/* ---- FusedLIDAR stacked observation (level5; fused_lidar.py:73-109,223-326, lidar_buffer.py:10-157) ----
 * own sphere + up to n_neighbors_max re-projected neighbour spheres, padded to TE_STACK_SPHERES, shuffled. */
#define TE_RING_DEPTH 10        /* snapshots kept per wingman (base_lidar.py:40: max_buffer_size=10) */
#define TE_STACK_NEIGHBORS_MIN 1
#define TE_STACK_NEIGHBORS_MAX 5 /* n ~ random.choice(range(1, 5)) = 1..4 neighbours (fused_lidar.py:77) */
#define TE_STACK_SPHERES 6      /* n_neighbors_max + 1 (fused_lidar.py:305) */
#define TE_OBS_STACKED_WORDS (TE_STACK_SPHERES * TE_OBS_LIDAR_WORDS) /* 6084 floats = 24 336 B per env */
/* one ring entry = one wingman's PerceptionSnapshot of one step: 12 header words + 4 per kept feature */
#define TE_RING_HEADER_WORDS 12 /* [0] i32 step stamp (0 = empty)  [1] i32 n_features  [2..4] position  [5..8] quaternion xyzw  [9..11] 0 */
#define TE_RING_FEATURE_WORDS 4 /* r_hat, theta, phi (own frame of the publisher), i32 entity type | publisher slot << 8 */
#define TE_RING_ENTRY_WORDS(D) (TE_RING_HEADER_WORDS + TE_RING_FEATURE_WORDS * ((D) - 1))

/* ---- quadrotor model: PyFlyt 0.11.1 QuadX "cf2x" + Bullet free body ------
 * The sources of these numbers (pyflyt 0.11.1, pybullet 3.2.7; poetry.lock:1412-1440)
 * are NOT in the reference tree; te_config_default() fills the table recorded in
 * SURVEY.md Appendix B (UNVERIFIED) so it can be corrected in one place. */
typedef struct te_quad_params {
  float mass;          /* kg */
  float inertia[3];    /* diag(Ixx, Iyy, Izz) kg m^2 */
  float arm;           /* |x| = |y| of each propeller, m */
  float total_thrust;  /* N at throttle 1 on all four motors */
  float thrust_coef;   /* N / rpm^2 */
  float torque_coef;   /* N m / rpm^2 */
  float motor_tau;     /* s, first-order lag */
  float noise_ratio;   /* multiplicative Gaussian throttle noise */
  float drag_coef_xyz, drag_area_xyz, drag_coef_pqr;
  float air_density;
  float gravity;       /* m/s^2, applied along -z (level4_simulation.py:69-70) */
  float ang_vel_kp[3], ang_vel_ki[3], ang_vel_kd[3], ang_vel_lim[3];
  float ang_pos_kp[3], ang_pos_lim[3];   /* ki = kd = 0 in cf2x: loop is memory-less */
  float lin_vel_kp[2], lin_vel_ki[2], lin_vel_kd[2], lin_vel_lim[2];
  float lin_pos_kp[2], lin_pos_lim[2];   /* mode 7 only; ki = kd = 0 */
  float z_pos_kp, z_pos_lim;             /* mode 7 only; ki = kd = 0 */
  float z_vel_kp, z_vel_ki, z_vel_kd, z_vel_lim;
  float pwm_floor;     /* 0.05: minimum pwm after saturation handling */
} te_quad_params;

/* ---- configuration ----------------------------- */
typedef struct te_config {
  uint32_t struct_size;   /* = sizeof(te_config); checked by te_create */
  int32_t task;           /* TE_TASK_* */
  int32_t n_envs;         /* environments owned by this te_env (this GPU's shard) */
  int32_t n_pursuers;     /* P; slot 0 is the RL agent */
  int32_t n_invaders;     /* I */
  int64_t env_index_base; /* global index of local env 0: RNG is keyed on the GLOBAL env index, so
                             results do not depend on how envs are sharded over GPUs */
  uint64_t seed;

  float dome_radius;      /* 20 (level4), 10 (stage01), 8 (stage02) */
  float lidar_radius;     /* 2 * dome_radius */
  float max_speed;        /* 10 km/h = 2.7778 m/s: observation normaliser only (quadcopter.py:590-600) */

  int32_t substeps;       /* physics sub-steps per env.step: 8 sim steps x 2 (level4_simulation.py:84-98) = 16.
                             0 (with observe_lag = 0) = no physics: the step's task logic alone, on the state as loaded */
  float physics_dt;       /* 1/240 */
  float control_dt;       /* 1/120: PID period, although update_control runs every sub-step (reference quirk) */
  int32_t observe_lag;    /* 1: IMU is read before the last integration */

  float shoot_range, explosion_range, origin_range, hit_prob;
  int32_t cooldown_steps; /* 4 s / (1/15 s) = 60 (gun.py:25) */
  int32_t munition;       /* per pursuer (20 in exp02/03; 4 for the stage02 agent; 0 in stage01) */
  int32_t max_step;       /* 300 (level4, stage01), 600 (stage02) */
  int32_t step_increment; /* +100 per step with >= 1 successful shot (exp03_vFinal_task.py:150-153) */
  int32_t n_rounds;       /* waves; = calculate_rounds(P, munition) (exp03_vFinal_task.py:198-226) */
  float born_radius;      /* 6 */
  float born_min_z;       /* 4 */
  float pursuer_spawn_radius; /* 2 (level4), 1 (stage02) */
  float invader_speed;    /* 0.4 kamikaze (…air_combat_only.py:58); 0.5 hover magnitude in stage02 */
  float ally_speed;       /* 0.6 (loyalwingman_navigator.py:37) */
  int32_t ally_policy;    /* TE_ALLY_* */
  float approach_bonus_gain; /* 1 (exp02/03), 10 (exp04, stage01, stage02) */
  float catch_distance;   /* stage01: 0.4 */
  float building_position[3]; /* (0,0,0.1) */

  int32_t motor_noise;    /* 0/1 */
  int32_t auto_reset;     /* 1: VecEnv semantics (reset inside step when done) */
  int32_t kamikaze_cone_check; /* 0: air-combat-only navigator (_is_building_path_clear == False, used by every
                                  vFinal task); 1: cone test of loitering_munition_navigator.py:78-87 */
  int32_t stacked_obs;    /* 1: keep the per-wingman snapshot ring and serve te_step_stacked (level5) */
  int32_t evaluation;     /* bit 0 (TE_EVAL_ON) = 1: Evaluation_Task rules (tasks/evaluation_task.py): EVERY pursuer is flown by the behaviour tree
                             (drivers of type "bt", :257-275,655-668; te_step's actions are ignored), reward 0 (:508-515), no
                             invaders-in-origin rule (:397), termination = time limit only if max_step > 0 (TIME_IS_LIMITED,
                             :519-524), all rounds over, anybody outside the dome, all pursuers destroyed (:526-551); kills are
                             counted per wingman (TE_D_KILLS) for te_wingman_info.  Bits 8.. : mask of the pursuers whose driver
                             is the CALLER's instead of the behaviour tree (drivers of type "nn", :264-268): bit 8 + p set =
                             pursuer p is observed with te_observe_wingman and commanded with te_set_wingman_actions 
                             Bit 1 (TE_EVAL_ORIGIN_RULE): invaders that reach the origin are still removed (Level52BTEvaluationTask.on_step_middle,
                             level5_2bt_evaluation_task.py:328; Evaluation_Task has that call commented out). */
  int32_t ground_contact; /* 1: ground plane (OPT-IN, off in every preset; parity unpinned): the reference loads plane.urdf at
                             z = -6 (entities_manager.py:120-124, immovable_structures.py:123-135); a drone whose hull bottom
                             reaches it stops there: inelastic normal contact (Bullet's default restitution 0) and Coulomb
                             friction 0.5 on the tangential velocity; no contact torque, no drone-drone contact */
  float ground_z;         /* -6 */
  float hull_half_height; /* 0.0125: half of the cf2x collision cylinder's height (unverified) */

  /* ---- SURVEY.md Appendix A.7 switches (ABI 4) ---- */
  int32_t control_every_substep; /* 1 (every preset): QuadX.update_control runs on EVERY physics sub-step, as the reference's
                             simulation loop calls it (level4_simulation.py:92-94), with the PID period still control_dt;
                             0: PyFlyt-native, update_control on every (physics_hz / ctrl_hz)-th sub-step only (120 Hz), the
                             motors keep the last pwm in between */
  int32_t lidar_channels; /* 3 (distance, flag, time: FusedLIDAR own sphere); 2 = the legacy LIDAR's layout
                             (distance, flag) that the level2-4 docs and the level5 teacher space quote
                             (sensors/lidar.py:137-145, level5_envrionment.py:79-84): obs_lidar is [N,2,13,26] */
  int32_t io_location;    /* TE_IO_DEVICE (0): every I/O pointer is a device pointer.  TE_IO_HOST (1): the actions, observation,
                             reward, done, info and terminal pointers of te_step / te_observe and te_reset's mask are HOST
                             pointers (what a SubprocVecEnv caller holds); the library stages them through device buffers of
                             its own with hipMemcpyAsync on the call's stream and the call returns when they have landed */
  int32_t drone_contact;  /* 1: drone-drone contact (OPT-IN, off in every preset; parity unpinned): armed drones collide as
                             spheres of contact_radius (Bullet narrow phase with the cf2x collision shapes; group/mask 1/1
                             for armed drones only, quadcopter.py:484-496): inelastic impulse along the line of centres at
                             the end of each sub-step, equal masses, no friction, no contact torque */
  float contact_radius;   /* 0.06: radius of the cf2x collision cylinder (unverified) */
  /* ---- round rule and the switches Level5DumbMultiObjectTask needs (ABI 5) ---- */
  int32_t initial_invaders;   /* 1: invaders armed in round r = min((r - 1) * invaders_per_round + initial_invaders, n_invaders): */
  int32_t invaders_per_round; /* 1:   Task.setup_round arms `round` of them (exp03_vFinal_task.py:180-196); the dumb multi-object task
                                 starts with 5 (level5_dumb_multiobject_task.py:87-90,173-184) */
  int32_t agent_scripted;     /* 1: pursuer 0 obeys the behaviour tree like the other wingmen and te_step's actions are ignored
                                 (level5_dumb_multiobject_task.py:256-266; implied by cfg.evaluation) */
  int32_t reward_model;       /* TE_REWARD_EXP03 (0): exp03_vFinal_task.py:423-515; TE_REWARD_L5_DUMB (1): level5_dumb_multiobject_task.py:452-553
                                 (reload-distance shaping, weighted kills / deaths, bounded border term, clipped to +-3000) 
                                 TE_REWARD_L5_C1 (2): Level5C1FusionTask.compute_reward (level5_c1_fusion_task.py:448-485): 10 |v| while the agent is closer to
                                 ITS closest invader than at the FIRST reward of this te_env's env (`self.last_distance` is set once and never
                                 again, not even by a reset: TE_E_LAST_DIST holds it, 0 = not measured yet), + 1000 per agent kill, - 2000 for
                                 the agent's suicide, clipped to +-3000 */
  int32_t agent_death_terminates; /* 1: the episode ends when pursuer 0 is disarmed (training tasks, exp03_vFinal_task.py:556-561);
                                 0: it goes on (level5_dumb_multiobject_task.py:600-606 has the test commented out) */
  int32_t quad_preset;    /* which table te_config_default / te_quad_preset filled `quad` with: TE_QUAD_CF2X_RECALLED (0) or
                             TE_QUAD_CF2X_RECORDED_FIT (1).  Informational: the kernels read `quad` only */

  te_quad_params quad;
} te_config;

enum { TE_IO_DEVICE = 0, TE_IO_HOST = 1 };
enum { TE_REWARD_EXP03 = 0, TE_REWARD_L5_DUMB = 1, TE_REWARD_L5_C1 = 2 };
/* quadrotor parameter presets (te_quad_preset).  RECALLED = PyFlyt 0.11.1's cf2x.yaml / cf2x.urdf as recalled in SURVEY.md
 * Appendix B (the default until round 2; chi^2 / dof 5.3 against the recording below); RECORDED_FIT = the same table with the
 * fewest changed entries (ang_vel_kp roll / pitch x 6, motor_tau x 0.4) that reproduce the only PyBullet-made numbers in the
 * reference tree (io_data0.h5: seven wingmen, two steps after a respawn) within their motor-noise scatter (chi^2 / dof 0.29) —
 * THE DEFAULT OF EVERY TASK since round 3; see DESIGN.md 5 and tools/physics_fit.py.  Neither is verified against the PyFlyt sources, and
 * the recording does not single the fit out: scored in units of its own (smaller) motor-noise scatter it reaches chi^2 / dof 2.8, and other
 * candidates (e.g. 120 Hz control with ang_pos_kp x 2 and total_thrust x 2) fit the 147 numbers as well or better.  One of several tables
 * the data support; parity with PyBullet is unpinned under either preset. */
enum { TE_QUAD_CF2X_RECALLED = 0, TE_QUAD_CF2X_RECORDED_FIT = 1 };

/* ---- state blob (te_get_state / te_set_state) ----------------------------
 * Env-major array of 4-byte words: for env e,
 *   drone d (0..D-1, pursuers first): TE_DRONE_WORDS words at (e*D + d)*TE_DRONE_WORDS
 *   then, after all N*D drone records, env record e: TE_ENV_WORDS words.
 * Words are float32 unless marked i32 (raw two's complement int32). */
enum {
  TE_D_POS = 0,        /* 3  world position */
  TE_D_QUAT = 3,       /* 4  x,y,z,w (Bullet convention) */
  TE_D_VEL = 7,        /* 3  world linear velocity */
  TE_D_OMEGA = 10,     /* 3  world angular velocity */
  TE_D_THROTTLE = 13,  /* 4  motor throttles */
  TE_D_PID_AV_I = 17,  /* 3  angular-velocity PID integral */
  TE_D_PID_AV_E = 20,  /* 3  angular-velocity PID previous error */
  TE_D_PID_LV_I = 23,  /* 2  linear-velocity PID integral */
  TE_D_PID_LV_E = 25,  /* 2  linear-velocity PID previous error */
  TE_D_PID_ZV_I = 27,  /* 1 */
  TE_D_PID_ZV_E = 28,  /* 1 */
  TE_D_SETPOINT = 29,  /* 4  [vx, vy, vr, vz] (mode 6) or [x, y, r, z] (mode 7) */
  TE_D_OBS_POS = 33,   /* 3  last IMU read: position (imu.py:27-41) */
  TE_D_OBS_EULER = 36, /* 3  roll, pitch, yaw */
  TE_D_OBS_VEL = 39,   /* 3  body-frame linear velocity */
  TE_D_OBS_RATE = 42,  /* 3  body-frame angular velocity */
  TE_D_FORMATION = 45, /* 3  last replace() position (quadcopter.py:437) */
  TE_D_PENDING = 48,   /* 6  world force(3)+torque(3) applied outside the loop, consumed by the next
                             integration (stage01 replace_invader, level2/components/quadcopter_manager.py:166-179) */
  TE_D_ALLY_ACTION = 48, /* 4  the same words of a CALLER-DRIVEN pursuer (pursuer 1 under TE_ALLY_EXTERNAL, the pursuers of
                             cfg.evaluation's driver mask; no level4 drone has a pending wrench):
                             the ally policy's previous action, Exp05_vFinal_Task.last_action (exp05_vFinal_task.py:139,259) */
  TE_D_ARMED = 54,     /* i32 */
  TE_D_MUNITION = 55,  /* i32 */
  TE_D_LAST_FIRED = 56,/* i32 */
  TE_D_NAV_STATE = 57, /* i32 TE_NAV_* (invaders) */
  TE_D_KILLS = 57,     /* i32 the same word of a PURSUER under cfg.evaluation: Evaluation_Task.lw_kills of this episode
                             (evaluation_task.py:498-499); 0 otherwise */
  TE_DRONE_WORDS = 58
};
enum {
  TE_E_STEP = 0,         /* i32 RL step counter of the episode */
  TE_E_MAX_STEP = 1,     /* i32 */
  TE_E_ROUND = 2,        /* i32 current wave */
  TE_E_LAST_DIST = 3,    /* f32 last_closest_distance (level4) / last_distance (stage01) */
  TE_E_AGENT_KILLS = 4,  /* i32 */
  TE_E_ALLIES_KILLS = 5, /* i32 */
  TE_E_DEADS = 6,        /* i32 */
  TE_E_SNAP_MASK = 7,    /* i32 bit d = drone d was armed when the offsets were last computed
                                (level4/components/entities_management/offsets_handler.py:68-95) */
  TE_E_EPISODE = 8,      /* i32 episode counter (RNG stream selector) */
  TE_E_LAST_ACTION = 9,  /* 4 f32 */
  TE_E_PREV_SNAP_MIN = 13, /* f32 stage02: last_closest_pursuer_to_invader_distance */
  TE_E_INFO_WAVE = 15,   /* i32 the wave as the step's compute_info saw it: BEFORE on_step_end may have started the next one (the environments read
                            info between on_step_middle and on_step_end, evaluation_environment.py:178-186); te_wingman_info reports this */
  TE_E_SNAP_MASK_HI = 14,/* i32 bits 32..63 of the snapshot mask (more than 32 drones per env: TE_TASK_LEVEL5_DUMB has 37) */
  TE_ENV_WORDS = 16
};

typedef struct te_env te_env; /* opaque */

/* Fill `cfg` with the reference constants of `task` (n_envs = 1, seed = 0).
 * Mirrors Task.init_constants (exp03_vFinal_task.py:88-112, level3/components/stages.py:65-83,
 * level2/pyflyt_level2_environment_modified_v2.py:27-47). */
int te_config_default(te_config* cfg, int32_t task);

/* Overwrite cfg->quad (and cfg->quad_preset) with one of the TE_QUAD_* tables. */
int te_quad_preset(te_config* cfg, int32_t preset);

/* Task.calculate_rounds (exp03_vFinal_task.py:198-226): the number of waves (= invader slots) `defenders` pursuers with
 * `munition` rounds each can clear; what te_config_default puts into n_rounds / n_invaders. */
int te_calculate_rounds(int32_t defenders, int32_t munition);

/* Replaces N x `Env.__init__` (exp03_vFinal_environment.py:42-63): allocates HBM state for
 * cfg->n_envs environments on `device_id` and runs on_env_init + on_episode_start. */
int te_create(const te_config* cfg, int32_t device_id, te_env** out);
void te_destroy(te_env* env);

/* Replaces `Env.reset` (exp03_vFinal_environment.py:128-146) for every env whose
 * env_mask byte is non-zero (NULL = all).  env_mask is a device pointer of n_envs bytes. */
int te_reset(te_env* env, const uint8_t* env_mask, void* stream);

/* Replaces `Env.compute_observation` (exp03_vFinal_environment.py:200-228) without stepping:
 * obs_lidar [N,3,13,26], obs_inertial [N,15], obs_last_action [N,4] (float32, device).
 * Immediately after te_create/te_reset the LIDAR plane is the empty sphere (all ones). */
int te_observe(te_env* env, float* obs_lidar, float* obs_inertial, float* obs_last_action, void* stream);

/* Replaces `VecEnv.step_async + step_wait` over N x `Env.step`
 * (exp03_vFinal_environment.py:150-171; SB3 auto-reset when cfg.auto_reset).
 *   actions        [N,4] float32 (direction xyz in [-1,1], magnitude in [0,1])
 *   reward [N] f32, done [N] u8, info [N,4] i32
 *   terminal_*     may be NULL; rows are written ONLY for envs with done != 0
 *                  (SB3 infos[i]["terminal_observation"]). */
int te_step(te_env* env, const float* actions, float* obs_lidar, float* obs_inertial,
            float* obs_last_action, float* reward, uint8_t* done, int32_t* info,
            float* terminal_lidar, float* terminal_inertial, float* terminal_last_action,
            void* stream);

/* Level5 (cfg.stacked_obs = 1): the same step with the FusedLIDAR stacked observation of the agent instead of
 * its own sphere (threatsense/level5/level5_envrionment.py:312-351; fused_lidar.py:73-109,223-326):
 *   obs_stacked [N,6,3,13,26] f32 — own sphere, 1..4 re-projected neighbour snapshots of random age (farther
 *                wins), padded with empty spheres, then shuffled;   obs_mask [N,6] u8 — 1 = a valid sphere.
 * Every armed wingman's snapshot (pose + the features of its own sphere) is pushed into a 10-deep ring per step;
 * the ring is part of the state blob.  terminal_* as in te_step.  te_observe_stacked is te_observe's counterpart
 * (after a reset every sphere is empty and every mask byte 0: there is no snapshot yet).  Both serve up to 37 drones per env (P <= 7;
 * TE_TASK_LEVEL5_FUSION has 36); te_observe, the own-sphere observation, stops at 32.  Three launches after the sub-step kernel: the engage
 * kernel, then one wave per (64-env chunk, wingman) pushing this step's ring entries, then one 5-wave workgroup per chunk for the
 * observer's view (te_stackview.hpp). */
int te_step_stacked(te_env* env, const float* actions, float* obs_stacked, uint8_t* obs_mask, float* obs_inertial,
                    float* obs_last_action, float* reward, uint8_t* done, int32_t* info, float* terminal_stacked,
                    uint8_t* terminal_mask, float* terminal_inertial, float* terminal_last_action, void* stream);
int te_observe_stacked(te_env* env, float* obs_stacked, uint8_t* obs_mask, float* obs_inertial, float* obs_last_action,
                       void* stream);

/* Level5DumbMultiObs.step + compute_info (threatsense/level5/level5_dumb_multiobs.py:116-150): one env.step in which every wingman is
 * scripted, returning the STUDENT observation of every pursuer and the teacher's action for it (the collector keeps the rows of the armed
 * ones, apps/threatsense_runner/collect_and_save.py:99-127):
 *   stacked [N,P,6,3,13,26] f32, mask [N,P,6] u8, inertial [N,P,15] f32 (normalised IMU + gun state of pursuer p),
 *   last_action [N,P,4] f32 = the behaviour tree's command of this step (direction, magnitude: pursuer.last_action = the teacher's
 *   action), active [N,P] u8 = pursuer p is armed after the step (its row is one the reference emits); reward / done / info as te_step.
 * Needs cfg.stacked_obs and cfg.agent_scripted (or cfg.evaluation).  Envs that auto-reset come back with the reset observation (empty
 * spheres, zero masks); there are no terminal buffers (the collector has no use for them). */
int te_step_students(te_env* env, float* stacked, uint8_t* mask, float* inertial, float* last_action, uint8_t* active, float* reward,
                     uint8_t* done, int32_t* info, void* stream);

/* exp05 (cfg.ally_policy == TE_ALLY_EXTERNAL): the two halves of Exp05_vFinal_Task.drive_lw_rl_agent
 * (exp05_vFinal_task.py:252-260), which the reference runs in on_step_start, i.e. BEFORE the physics of a step:
 *   te_observe_ally      = compute_lw_observation (:265-292) of pursuer 1 on the CURRENT state: its own LIDAR sphere
 *                          ally_lidar [N,3,13,26], normalised inertial data + gun state ally_inertial [N,15], the ally
 *                          policy's previous action ally_last_action [N,4] (zeros after a reset); ally_active [N] u8 = 1
 *                          where the reference would call driver.predict (the ally is armed).  Any pointer may be NULL.
 *   te_set_ally_actions  = pursuer.drive(action) for every env whose ally is armed: ally_actions [N,4] f32 as te_step's
 *                          actions; remembered as the ally's last action.  Rows of envs with a dead ally are ignored.
 * One env.step of exp05 is therefore te_observe_ally -> (the caller's policy) -> te_set_ally_actions -> te_step. */
int te_observe_ally(te_env* env, float* ally_lidar, float* ally_inertial, float* ally_last_action, uint8_t* ally_active,
                    void* stream);
int te_set_ally_actions(te_env* env, const float* ally_actions, void* stream);
/* The same pair for ANY caller-driven pursuer `wingman`: pursuer 1 of exp05 (= the two calls above), or a pursuer whose bit
 * is set in cfg.evaluation's driver mask (Evaluation_Task.drive_lw with a driver that has `predict`,
 * evaluation_task.py:257-268, compute_lw_observation :281-310).  The last action is kept per wingman (the reference shares
 * one `self.last_action` between all "nn" drivers of an env, i.e. a wingman sees its predecessor's action: not restated). */
int te_observe_wingman(te_env* env, int32_t wingman, float* lidar, float* inertial, float* last_action, uint8_t* active, void* stream);
int te_set_wingman_actions(te_env* env, int32_t wingman, const float* actions, void* stream);

/* Evaluation_Task.compute_info (evaluation_task.py:553-574), cfg.evaluation only: wingman_info [N,P,5] i32 with the rows
 * (lw_kills, lw_alive, lw_munitions, current_wave, step) of every pursuer AFTER the last te_step (the reference lists the
 * armed pursuers only: a row with lw_alive == 0 is one it would omit).  For envs that auto-reset in that step the rows
 * describe the fresh episode; callers that want an episode's final rows read them with cfg.auto_reset = 0, as the
 * reference's evaluation loop does (evaluation_exp01_1bt_app_ready.py:80-96). */
int te_wingman_info(te_env* env, int32_t* wingman_info, void* stream);

/* Persistent observation (opt-in; no counterpart in the reference, whose sensors build a fresh numpy array every call,
 * fused_lidar.py:143-262).  on != 0: the caller promises that nobody but this library writes the LIDAR observation buffer it passes
 * (obs_lidar of te_step; obs_stacked of te_step_stacked / te_step_students / te_observe_stacked).  While the SAME pointer keeps coming, a
 * call rewrites only the cells that change (the cells the previous observation patched go back to 1, the new ones are patched) instead of
 * streaming the whole background; the buffer ends up bit for bit as the dense path leaves it.  A different pointer (e.g. the slots of a
 * rollout buffer) takes the dense path for that call.  Terminal buffers are always dense.  on == 0 (the default): every call is dense.
 * The algorithmic bytes of such a step are the cells it touches, not the background: bench.py keeps the dense path as its headline.
 * "The same buffer" is recognised by its ADDRESS alone: the caller must keep the allocation alive for as long as it passes it — a buffer
 * that is freed and whose memory is handed to a NEW allocation at the same address (a caching allocator does that) would be taken for the old
 * one and its background never written.  te_set_persistent_obs(env, 0) followed by (env, 1) forgets the remembered buffer: call it whenever
 * the buffer's identity changes.  Under cfg.io_location == TE_IO_HOST the persistent buffer is the library's own staging (the caller's host
 * array is filled from it every call); the terminal rows of done envs are compacted in a staging of their own. */
int te_set_persistent_obs(te_env* env, int32_t on);

/* Synthetic random-action generator of the throughput harness
 * (apps/threatengage_runner/interactive/analyse.py:55-59): dir ~ U(-1,1)^3, mag ~ U(0,1),
 * Philox4x32-10 keyed on (seed, global env index, step_index). */
int te_random_actions(te_env* env, float* actions, uint64_t seed, uint64_t step_index, void* stream);

/* Checkpoint / parity hooks (the reference never checkpoints env state). */
int te_state_words(const te_env* env, size_t* out_words);
int te_get_state(te_env* env, void* dst_device, size_t words, void* stream);
int te_set_state(te_env* env, const void* src_device, size_t words, void* stream);

/* Algorithmic HBM bytes one te_step moves per environment (SURVEY.md 8(d) formula). */
int te_algorithmic_bytes_per_env_step(const te_config* cfg, size_t* out_bytes);

/* Kernel timing with HIP events on the stream te_step launches on.  After te_profile_begin the next
 * `max_steps` te_step calls launch their kernels with start / stop events (hipExtLaunchKernelGGL: the
 * dispatch's own begin / end timestamps, the quantity rocprofv3 --kernel-trace reports; TE_PROF=markers
 * in the environment selects marker packets between the launches instead); te_profile_end blocks until
 * they have finished and returns the average duration of the sub-step kernel and of the rest of the
 * step (engage kernel start to the end of the step's last kernel) in milliseconds, and how many steps
 * were recorded. */
int te_profile_begin(te_env* env, int32_t max_steps);
int te_profile_end(te_env* env, float* substeps_ms, float* engage_observe_ms, int32_t* n_steps);

/* Diagnostic builds only (-DTE_DEBUG_STAMPS): 100 MHz phase stamps of one workgroup of the last te_step;
 * zeros otherwise.  out_host is HOST memory. */
int te_debug_stamps(te_env* env, uint64_t* out_host, int32_t n);

int te_abi_version(void);
const char* te_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* THREATENGAGE_H */
