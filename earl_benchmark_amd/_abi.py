"""ctypes view of include/earl_tabletop.h and the loader of the HIP shared library.

The product path is GPU-only: if `csrc/libearl_hip.so` is missing or cannot be loaded this module
raises -- there is no CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never
imported from here).
"""
import ctypes as C
import os

EARL_OK = 0
REWARD_SPARSE, REWARD_DENSE = 0, 1
REWARD_TYPES = {'sparse': REWARD_SPARSE, 'dense': REWARD_DENSE}


class TabletopCfg(C.Structure):
  _fields_ = [('n', C.c_int32), ('env_offset', C.c_int32), ('reward_type', C.c_int32), ('wide_init', C.c_int32),
              ('reset_at_goal', C.c_int32), ('horizon', C.c_int32), ('goal_change_frequency', C.c_int32),
              ('auto_reset', C.c_int32), ('n_goals', C.c_int32), ('n_sample_goals', C.c_int32),
              ('seed', C.c_uint64), ('counter', C.c_uint64)]


class TabletopState(C.Structure):
  _fields_ = [('qpos', C.c_void_p), ('attached', C.c_void_p), ('goal_idx', C.c_void_p), ('goal_table', C.c_void_p),
              ('steps_since_reset', C.c_void_p), ('num_interventions', C.c_void_p),
              ('steps_since_goal_change', C.c_void_p), ('lifelong_return', C.c_void_p), ('counter_base', C.c_void_p)]


class TabletopOut(C.Structure):
  _fields_ = [('obs', C.c_void_p), ('reward', C.c_void_p), ('done', C.c_void_p), ('success', C.c_void_p), ('reward_f64', C.c_void_p)]


class MotorParams(C.Structure):   # struct earl_motor_params (include/earl_glue.h)
  _fields_ = [('kp', C.c_double), ('kd', C.c_double), ('voltage', C.c_double), ('viscous_damping', C.c_double),
              ('torque_control', C.c_int32)]


class KitchenParams(C.Structure):   # struct earl_kitchen_params (include/earl_glue.h)
  _fields_ = [('pos_bound', C.c_double * 2 * 23), ('vel_bound', C.c_double * 2 * 23), ('pos_noise_amp', C.c_double * 23),
              ('act_mid', C.c_double * 9), ('act_amp', C.c_double * 9), ('mocap_range', C.c_double * 3),
              ('mocap_clip_lower', C.c_double * 3), ('mocap_clip_upper', C.c_double * 3), ('step_duration', C.c_double),
              ('robot_noise_ratio', C.c_double)]


class SawyerCfg(C.Structure):   # struct earl_sawyer_cfg (include/earl_physics.h)
  _fields_ = [('n', C.c_int32), ('env_offset', C.c_int32), ('reward_type', C.c_int32), ('horizon', C.c_int32),
              ('frame_skip', C.c_int32), ('att_hand', C.c_int32), ('att_right', C.c_int32), ('att_left', C.c_int32),
              ('att_obj', C.c_int32), ('obj_dof', C.c_int32), ('obj_kind', C.c_int32), ('n_goal_rows', C.c_int32), ('goal_change_frequency', C.c_int32), ('n_wide', C.c_int32),
              ('att_grasp', C.c_int32), ('att_lpad', C.c_int32), ('att_rpad', C.c_int32), ('pad2_', C.c_int32),
              ('action_scale', C.c_double),
              ('mocap_low', C.c_double * 3), ('mocap_high', C.c_double * 3), ('mocap_quat', C.c_double * 4),
              ('success_radius', C.c_double), ('hand_init_pos', C.c_double * 3), ('obj_init_pos', C.c_double * 3),
              ('obj_init_angle', C.c_double), ('angle_noise', C.c_double * 2),
              ('obj_low', C.c_double * 3), ('obj_high', C.c_double * 3), ('obj_reject_xy', C.c_double * 2), ('obj_reject_radius', C.c_double),
              ('goal_table', C.c_void_p), ('wide_table', C.c_void_p), ('wide_shift', C.c_double * 3), ('wide_noise', C.c_double),
              ('init_tcp', C.c_double * 3), ('box_corners', C.c_double * 3 * 4),
              ('seed', C.c_uint64), ('counter', C.c_uint64), ('step_counter', C.c_uint64)]


class SawyerState(C.Structure):
  _fields_ = [('qpos', C.c_void_p), ('qvel', C.c_void_p), ('mocap_pos', C.c_void_p), ('goal', C.c_void_p),
              ('steps_since_reset', C.c_void_p), ('steps_since_goal_change', C.c_void_p), ('obj_init', C.c_void_p),
              ('last_obs', C.c_void_p), ('fail_count', C.c_void_p), ('sched', C.c_void_p)]


class SawyerOut(C.Structure):
  _fields_ = [('obs', C.c_void_p), ('reward', C.c_void_p), ('done', C.c_void_p), ('success', C.c_void_p), ('status', C.c_void_p), ('info', C.c_void_p)]


SAWYER_INFO = 8       # EARL_SAWYER_INFO; slots EARL_INFO_* (include/earl_physics.h)
SAWYER_INFO_KEYS = ('success', 'near_object', 'grasp_success', 'grasp_reward', 'in_place_reward', 'obj_to_target', 'unscaled_reward')
STEP_DIVERGED = 1     # EARL_STEP_DIVERGED (include/earl_physics.h)


class KitchenCfg(C.Structure):     # struct earl_kitchen_cfg (include/earl_physics.h)
  _fields_ = [('n', C.c_int32), ('env_offset', C.c_int32), ('horizon', C.c_int32), ('frame_skip', C.c_int32), ('sensor_noise', C.c_int32),
              ('n_att', C.c_int32), ('site_att', C.c_int32 * 8), ('seed', C.c_uint64), ('counter', C.c_uint64), ('mocap_quat_dev', C.c_void_p)]


class KitchenState(C.Structure):   # struct earl_kitchen_state
  _fields_ = [(k, C.c_void_p) for k in ('qpos', 'qvel', 'mocap_pos', 'goal', 'last_qp_robot', 'att_xpos', 'steps_since_reset', 'fail_count', 'last_obs',
                                        'action64', 'ctrl9', 'noise', 'qpos_bak', 'qvel_bak', 'sites', 'bad', 'mocap_bak', 'att_bak')]


class KitchenOut(C.Structure):     # struct earl_kitchen_out
  _fields_ = [(k, C.c_void_p) for k in ('obs', 'reward', 'done', 'success', 'status')]


class MinitaurCfg(C.Structure):    # struct earl_minitaur_cfg (include/earl_physics.h)
  _fields_ = [('n', C.c_int32), ('env_offset', C.c_int32), ('horizon', C.c_int32), ('num_substeps', C.c_int32), ('settle_steps', C.c_int32),
              ('randomize', C.c_int32), ('n_goals', C.c_int32), ('goal_change_frequency', C.c_int32), ('overheat_steps', C.c_int32),
              ('motor_dof', C.c_int32 * 8), ('pad_', C.c_int32), ('motor_dir', C.c_double * 8), ('motor_kp', C.c_double), ('motor_kd', C.c_double),
              ('motor_velocity_limit', C.c_double), ('overheat_torque', C.c_double), ('distance_weight', C.c_double), ('energy_weight', C.c_double),
              ('success_radius', C.c_double), ('goal_table', C.c_void_p), ('reset_qpos', C.c_void_p),
              ('base_mass_err', C.c_double * 2), ('leg_mass_err', C.c_double * 2), ('leg_mass', C.c_double), ('motor_mass', C.c_double),
              ('foot_friction', C.c_double * 2), ('seed', C.c_uint64), ('counter', C.c_uint64), ('step_counter', C.c_uint64)]


class MinitaurState(C.Structure):  # struct earl_minitaur_state
  _fields_ = [(k, C.c_void_p) for k in ('qpos', 'qvel', 'goal', 'motor_param', 'observed_torque', 'overheat', 'motor_enabled', 'steps_since_reset',
                                        'steps_since_goal_change', 'fail_count', 'last_obs')]


class MinitaurOut(C.Structure):    # struct earl_minitaur_out
  _fields_ = [(k, C.c_void_p) for k in ('obs', 'reward', 'done', 'success', 'status')]


_P = C.POINTER
# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/earl_tabletop.h one to one
SIGNATURES = {
    'earl_tabletop_step': [_P(TabletopCfg), _P(TabletopState), C.c_void_p, C.c_void_p, _P(TabletopOut), C.c_void_p],
    'earl_tabletop_rollout': [_P(TabletopCfg), _P(TabletopState), C.c_int32, C.c_void_p, _P(TabletopOut), C.c_void_p],
    'earl_tabletop_reset_rollout': [_P(TabletopCfg), _P(TabletopState), C.c_int32, C.c_void_p, _P(TabletopOut), C.c_void_p],
    'earl_tabletop_eval_episodes': [_P(TabletopCfg), _P(TabletopState), C.c_int32, C.c_int32, C.c_void_p, C.c_int64, _P(TabletopOut), C.c_void_p],
    'earl_tabletop_reset': [_P(TabletopCfg), _P(TabletopState), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_tabletop_observe': [_P(TabletopCfg), _P(TabletopState), _P(TabletopOut), C.c_void_p],
    'earl_tabletop_reward': [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_tabletop_valid_init': [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_tabletop3_step': [_P(TabletopCfg), _P(TabletopState), C.c_void_p, _P(TabletopOut), C.c_void_p],
    'earl_tabletop3_rollout': [_P(TabletopCfg), _P(TabletopState), C.c_int32, C.c_void_p, _P(TabletopOut), C.c_void_p],
    'earl_tabletop3_reset': [_P(TabletopCfg), _P(TabletopState), C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_tabletop3_reward': [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_debug_set_rollout_impl': [C.c_int],
    'earl_debug_set_rollout_wgs_per_cu': [C.c_int],
    'earl_debug_read_ws_profile': [C.c_void_p, C.c_int32],
    # include/earl_glue.h
    'earl_sawyer_sparse_f64': [C.c_int32, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_sawyer_sparse_f32': [C.c_int32, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_minitaur_leg_to_motor': [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_minitaur_motor_torque': [C.c_int32, _P(MotorParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_minitaur_reward': [C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_kitchen_reward': [C.c_int32] + [C.c_void_p] * 6,
    'earl_kitchen_default_params': [_P(KitchenParams)],
    'earl_kitchen_action': [C.c_int32, _P(KitchenParams)] + [C.c_void_p] * 5,
    'earl_kitchen_obs': [C.c_int32, _P(KitchenParams)] + [C.c_void_p] * 5,
    'earl_philox_uniform': [C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_uint32, C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    # include/earl_physics.h
    'earl_physics_step': [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7,
    'earl_physics_forward': [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 9,
    'earl_physics_model_size': [],
    'earl_physics_model24_size': [],
    'earl_collision_model_size': [],
    'earl_sawyer_cfg_size': [],
    'earl_physics_forget_table': [C.c_void_p],
    'earl_debug_set_physics_lanes': [C.c_int],
    'earl_debug_set_door_variant': [C.c_int],
    'earl_debug_set_peg_schedule': [C.c_int],
    'earl_kitchen_step': [C.c_void_p, C.c_void_p, _P(KitchenParams), _P(KitchenCfg), _P(KitchenState), C.c_void_p, _P(KitchenOut), C.c_void_p],
    'earl_kitchen_rollout': [C.c_void_p, C.c_void_p, _P(KitchenParams), _P(KitchenCfg), _P(KitchenState), C.c_void_p, C.c_int32, _P(KitchenOut), C.c_void_p],
    'earl_sawyer_rollout': [C.c_void_p, C.c_void_p, C.c_int32, _P(SawyerCfg), _P(SawyerState), C.c_void_p, C.c_int32, _P(SawyerOut), C.c_void_p],
    'earl_minitaur_rollout': [C.c_void_p, C.c_void_p, _P(MinitaurCfg), _P(MinitaurState), C.c_void_p, C.c_int32, _P(MinitaurOut), C.c_void_p],
    'earl_minitaur_reset': [C.c_void_p, C.c_void_p, _P(MinitaurCfg), _P(MinitaurState), C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_minitaur_cfg_size': [],
    'earl_debug_set_minitaur_stepper': [C.c_int],
    'earl_debug_set_minitaur_duo': [C.c_int],
    'earl_debug_set_solo': [C.c_int],
    'earl_debug_set_solo_mt': [C.c_int],
    'earl_sawyer_reset': [C.c_void_p, C.c_int32, _P(SawyerCfg), _P(SawyerState)] + [C.c_void_p] * 5,
    'earl_sawyer_observe': [C.c_void_p, C.c_int32, _P(SawyerCfg), _P(SawyerState), C.c_void_p, C.c_void_p],
    'earl_sawyer_door_reward': [_P(SawyerCfg), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_sawyer_door_info': [_P(SawyerCfg), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    'earl_version': [],
    'earl_last_error': [],
    'earl_device_count': [],
}
_RESTYPES = {'earl_version': C.c_char_p, 'earl_last_error': C.c_char_p}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libearl_hip.so')
_lib = None


class EarlHipError(RuntimeError):
  pass


def load():
  """Load csrc/libearl_hip.so (built by `__graft_entry__.build()` / `make -C earl_benchmark_amd/csrc`)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise EarlHipError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                         '(or `make -C earl_benchmark_amd/csrc`). There is no CPU fallback.')
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It must be the one
    # already mapped when our library's NEEDED libamdhip64.so.7 is resolved, otherwise the process ends up with two
    # HIP runtimes and kernels launched from here see "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
      fn = getattr(lib, name)  # AttributeError here = header/library mismatch
      fn.argtypes = argtypes
      fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
  return _lib


# ---- the host build of the tabletop per-env functions (csrc/tabletop_host.cpp -> csrc/libearl_host.so): the `_cpu` entry points of
# include/earl_tabletop.h.  Loaded ONLY when a caller asks for device='cpu' (BASELINE configs[0]: "1 env, CPU ... plumbing, no GPU");
# nothing falls back to it -- without a GPU the default device still raises EarlHipError.
HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libearl_host.so')
HOST_SIGNATURES = {name + '_cpu': argtypes[:-1] for name, argtypes in SIGNATURES.items()
                   if name.startswith(('earl_tabletop_', 'earl_tabletop3_'))}            # same arguments minus the stream
_host = None


class HostLib:
  """libearl_host.so behind the names (and argument lists) of the HIP library: `lib.earl_tabletop_step(..., stream)` calls
  `earl_tabletop_step_cpu(...)` -- host pointers, no stream -- so that envs/tabletop.py drives either with the same code."""

  def __init__(self, cdll):
    self._cdll = cdll
    for name, argtypes in HOST_SIGNATURES.items():
      fn = getattr(cdll, name)                 # AttributeError here = header/library mismatch
      fn.argtypes, fn.restype = argtypes, C.c_int
      setattr(self, name, fn)
      setattr(self, name[:-4], (lambda f: (lambda *a: f(*a[:-1])))(fn))
    cdll.earl_host_last_error.restype = C.c_char_p
    cdll.earl_host_version.restype = C.c_char_p
    cdll.earl_host_set_threads.argtypes = [C.c_int]
    self.earl_last_error = cdll.earl_host_last_error
    self.earl_version = cdll.earl_host_version
    self.set_threads = cdll.earl_host_set_threads


def load_host():
  """Load csrc/libearl_host.so (g++ build of csrc/tabletop_device.h + tabletop_step.h; `make -C earl_benchmark_amd/csrc`)."""
  global _host
  if _host is None:
    if not os.path.exists(HOST_LIB_PATH):
      raise EarlHipError(f'{HOST_LIB_PATH} not found: build it with `make -C earl_benchmark_amd/csrc libearl_host.so`')
    try:                                                                       # the library is compiled with -mavx2 -mfma: on a CPU without them it would die with SIGILL
      flags = next((ln for ln in open('/proc/cpuinfo') if ln.startswith('flags')), '')
    except OSError:
      flags = 'avx2 fma'                                                       # (no /proc: nothing to check against)
    missing = [f for f in ('avx2', 'fma') if f not in flags.split()]
    if missing:
      raise EarlHipError(f'{HOST_LIB_PATH} needs a CPU with {" and ".join(missing)} (built with -mavx2 -mfma); this one has neither the HIP path (device="cpu" was asked for) nor those')
    _host = HostLib(C.CDLL(HOST_LIB_PATH))
  return _host


def check(rc, what, lib=None):
  if rc != EARL_OK:
    msg = (lib or load()).earl_last_error()
    raise EarlHipError(f'{what} failed with code {rc}: {msg.decode() if msg else "?"}')
