"""ctypes mirror of include/threatengage.h (te_quad_params, te_config) and task constants.

The numbers themselves live in C (dronechase_amd/csrc/te_config.c: te_config_default), one table for
every consumer; this module only describes the struct layout to Python.
"""
from __future__ import annotations

import ctypes as C

TE_ABI_VERSION = 5

TASK_STAGE01, TASK_STAGE02, TASK_EXP02, TASK_EXP03, TASK_EXP04, TASK_LEVEL5, TASK_EXP05, TASK_EVALUATION, TASK_LEVEL5_DUMB, TASK_LEVEL5_2BT, TASK_LEVEL5_C1, TASK_LEVEL5_FUSION = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12
TASKS = {"stage01": TASK_STAGE01, "stage02": TASK_STAGE02, "exp02": TASK_EXP02, "exp03": TASK_EXP03,
         "stage03": TASK_EXP03, "exp04": TASK_EXP04, "level5": TASK_LEVEL5, "exp05": TASK_EXP05, "evaluation": TASK_EVALUATION,
         "level5_dumb": TASK_LEVEL5_DUMB, "level5_2bt": TASK_LEVEL5_2BT, "level5_c1": TASK_LEVEL5_C1, "level5_fusion": TASK_LEVEL5_FUSION}
REWARD_EXP03, REWARD_L5_DUMB, REWARD_L5_C1 = 0, 1, 2
EVAL_ON, EVAL_ORIGIN_RULE = 1, 2
ALLY_NONE, ALLY_BT, ALLY_FROZEN, ALLY_EXTERNAL = 0, 1, 2, 3
IO_DEVICE, IO_HOST = 0, 1
QUAD_CF2X_RECALLED, QUAD_CF2X_RECORDED_FIT = 0, 1

LIDAR_NTHETA, LIDAR_NPHI, LIDAR_CHANNELS = 13, 26, 3
LIDAR_CELLS = LIDAR_NTHETA * LIDAR_NPHI
OBS_LIDAR_WORDS = LIDAR_CHANNELS * LIDAR_CELLS
OBS_INERTIAL_WORDS = 15
OBS_ACTION_WORDS = 4
INFO_WORDS = 4
DRONE_WORDS = 58
ENV_WORDS = 16
# level5 stacked observation (TE_RING_* / TE_STACK_*)
RING_DEPTH, STACK_SPHERES = 10, 6
OBS_STACKED_WORDS = STACK_SPHERES * OBS_LIDAR_WORDS
RING_HEADER_WORDS, RING_FEATURE_WORDS = 12, 4


def ring_entry_words(n_drones: int) -> int:
    return RING_HEADER_WORDS + RING_FEATURE_WORDS * (n_drones - 1)

# word offsets inside a drone record of the state blob (TE_D_*)
D = dict(POS=0, QUAT=3, VEL=7, OMEGA=10, THROTTLE=13, PID_AV_I=17, PID_AV_E=20, PID_LV_I=23, PID_LV_E=25,
         PID_ZV_I=27, PID_ZV_E=28, SETPOINT=29, OBS_POS=33, OBS_EULER=36, OBS_VEL=39, OBS_RATE=42, FORMATION=45,
         PENDING=48, ALLY_ACTION=48, ARMED=54, MUNITION=55, LAST_FIRED=56, NAV_STATE=57, KILLS=57)
# word offsets inside an env record (TE_E_*)
E = dict(STEP=0, MAX_STEP=1, ROUND=2, LAST_DIST=3, AGENT_KILLS=4, ALLIES_KILLS=5, DEADS=6, SNAP_MASK=7, EPISODE=8,
         LAST_ACTION=9, PREV_SNAP_MIN=13, SNAP_MASK_HI=14, INFO_WAVE=15)
D_INT_WORDS = (54, 55, 56, 57)
E_INT_WORDS = (0, 1, 2, 4, 5, 6, 7, 8, 14, 15)


class QuadParams(C.Structure):
    _fields_ = [
        ("mass", C.c_float), ("inertia", C.c_float * 3), ("arm", C.c_float), ("total_thrust", C.c_float),
        ("thrust_coef", C.c_float), ("torque_coef", C.c_float), ("motor_tau", C.c_float), ("noise_ratio", C.c_float),
        ("drag_coef_xyz", C.c_float), ("drag_area_xyz", C.c_float), ("drag_coef_pqr", C.c_float),
        ("air_density", C.c_float), ("gravity", C.c_float),
        ("ang_vel_kp", C.c_float * 3), ("ang_vel_ki", C.c_float * 3), ("ang_vel_kd", C.c_float * 3),
        ("ang_vel_lim", C.c_float * 3),
        ("ang_pos_kp", C.c_float * 3), ("ang_pos_lim", C.c_float * 3),
        ("lin_vel_kp", C.c_float * 2), ("lin_vel_ki", C.c_float * 2), ("lin_vel_kd", C.c_float * 2),
        ("lin_vel_lim", C.c_float * 2),
        ("lin_pos_kp", C.c_float * 2), ("lin_pos_lim", C.c_float * 2),
        ("z_pos_kp", C.c_float), ("z_pos_lim", C.c_float),
        ("z_vel_kp", C.c_float), ("z_vel_ki", C.c_float), ("z_vel_kd", C.c_float), ("z_vel_lim", C.c_float),
        ("pwm_floor", C.c_float),
    ]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("task", C.c_int32), ("n_envs", C.c_int32), ("n_pursuers", C.c_int32),
        ("n_invaders", C.c_int32), ("env_index_base", C.c_int64), ("seed", C.c_uint64),
        ("dome_radius", C.c_float), ("lidar_radius", C.c_float), ("max_speed", C.c_float),
        ("substeps", C.c_int32), ("physics_dt", C.c_float), ("control_dt", C.c_float), ("observe_lag", C.c_int32),
        ("shoot_range", C.c_float), ("explosion_range", C.c_float), ("origin_range", C.c_float),
        ("hit_prob", C.c_float), ("cooldown_steps", C.c_int32), ("munition", C.c_int32), ("max_step", C.c_int32),
        ("step_increment", C.c_int32), ("n_rounds", C.c_int32), ("born_radius", C.c_float),
        ("born_min_z", C.c_float), ("pursuer_spawn_radius", C.c_float), ("invader_speed", C.c_float),
        ("ally_speed", C.c_float), ("ally_policy", C.c_int32), ("approach_bonus_gain", C.c_float),
        ("catch_distance", C.c_float), ("building_position", C.c_float * 3),
        ("motor_noise", C.c_int32), ("auto_reset", C.c_int32), ("kamikaze_cone_check", C.c_int32), ("stacked_obs", C.c_int32),
        ("evaluation", C.c_int32), ("ground_contact", C.c_int32), ("ground_z", C.c_float), ("hull_half_height", C.c_float),
        ("control_every_substep", C.c_int32), ("lidar_channels", C.c_int32), ("io_location", C.c_int32),
        ("drone_contact", C.c_int32), ("contact_radius", C.c_float),
        ("initial_invaders", C.c_int32), ("invaders_per_round", C.c_int32), ("agent_scripted", C.c_int32), ("reward_model", C.c_int32),
        ("agent_death_terminates", C.c_int32), ("quad_preset", C.c_int32),
        ("quad", QuadParams),
    ]

    @property
    def n_drones(self) -> int:
        return int(self.n_pursuers + self.n_invaders)

    def copy(self) -> "Config":
        out = Config()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(Config))
        return out


def apply_overrides(cfg: Config, **kw) -> Config:
    """Set plain or `quad__field` attributes; unknown names raise (no silent typos)."""
    for k, v in kw.items():
        if k.startswith("quad__"):
            name = k[len("quad__"):]
            if not hasattr(cfg.quad, name):
                raise AttributeError(f"te_quad_params has no field {name!r}")
            setattr(cfg.quad, name, v)
        else:
            if k not in {f[0] for f in Config._fields_}:
                raise AttributeError(f"te_config has no field {k!r}")
            setattr(cfg, k, v)
    return cfg
