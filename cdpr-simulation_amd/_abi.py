"""ctypes mirror of include/cdpr.h (the C-ABI of libcdpr_hip.so).

Only data definitions live here: the structs and constants of the boundary.  The
loader and the function prototypes are in `_native.py`.
"""
import ctypes as C

ABI_VERSION = 6
MAX_CABLES = 12
MAX_D_BUFFER = 32
MAX_D_DEGREE = 4
COMMAND_VELOCITY, COMMAND_POSITION, COMMAND_FORCE = 0, 1, 2  # cdpr_update_scheduled_kind
MAX_CASCADE = 4
PID_DEBUG_AXES = 9

OK = 0
IGNORED = 1
ERR_INVALID = -1
ERR_DEVICE = -2
ERR_UNSUPPORTED = -3
ERR_NOMEM = -4

STAGE_FK = 0x1
STAGE_TD = 0x2
STAGE_PID_DEBUG = 0x4

PLAN_FIRST_WORLD_STEP, PLAN_SCHEDULED, PLAN_ROLLOUT, PLAN_NOT_STEADY = 1, 2, 4, 8  # cdpr_plan_kernel flags
MAP_AUTO = 0
MAP_LANE_PER_ROBOT = 1
MAP_LANE_PAIR = 2
MAP_LANE_PER_CABLE = 3


class FilterParams(C.Structure):
    _fields_ = [("rel_cutoff", C.c_double), ("quality", C.c_double), ("cascade", C.c_uint32), ("reserved_", C.c_uint32)]


class PidParams(C.Structure):
    _fields_ = [
        ("forward_gain", C.c_double),
        ("p_gain", C.c_double),
        ("i_gain", C.c_double),
        ("d_gain", C.c_double),
        ("d_degree", C.c_uint32),
        ("d_buffer_length", C.c_uint32),
        ("i_limit", C.c_double),
        ("cmd_limit", C.c_double),
        ("p_filter", FilterParams),
        ("d_filter", FilterParams),
    ]


class ConfigStruct(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_cables", C.c_uint32),
        ("batch", C.c_uint64),
        ("dt", C.c_double),
        ("frame_anchor", (C.c_double * 3) * MAX_CABLES),
        ("platform_anchor", (C.c_double * 3) * MAX_CABLES),
        ("cable_ref_length", C.c_double * MAX_CABLES),
        ("home_pose", C.c_double * 7),
        ("mass", C.c_double),
        ("inertia", C.c_double * 6),
        ("gravity", C.c_double * 3),
        ("joint_damping", C.c_double),
        ("effort_limit", C.c_double),
        ("velocity_limit", C.c_double),
        ("unilateral_cables", C.c_uint32),
        ("precision", C.c_uint32),
        ("velocity_pid", PidParams),
        ("position_pid", PidParams),
        ("velocity_epsilon", C.c_double),
        ("publish_period", C.c_double),
        ("stages", C.c_uint32),
        ("mapping", C.c_uint32),
        ("fk_max_iterations", C.c_uint32),
        ("per_robot_commands", C.c_uint32),
        ("fk_lambda", C.c_double),
        ("fk_tolerance", C.c_double),
        ("td_f_min", C.c_double),
        ("td_f_max", C.c_double),
        ("passive_damping", C.c_double),
        ("leg_inertia", C.c_double),
        ("cable_axial_mass", C.c_double),
        ("anchor_point_mass", C.c_double),
        ("anchor_inertia", C.c_double),
        ("travel_lower", C.c_double),
        ("travel_upper", C.c_double),
        ("travel_stop", C.c_uint32),
        ("reserved3_", C.c_uint32),
    ]
