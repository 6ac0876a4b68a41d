"""ctypes binding of include/paintrl.h (the C ABI of libpaintrl_hip.so).

There is no CPU fallback: if the library is missing or does not export every
symbol of the header, importing fails loudly.
"""
import ctypes as C
import os

from . import build as _build

MAX_DISCRETE = 64
STATE_DOUBLES = 16
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint64)
_fp = C.POINTER(C.c_float)


class PrlPolicyWeights(C.Structure):
    _fields_ = [('in_dim', C.c_int32), ('h1', C.c_int32), ('h2', C.c_int32), ('n_actions', C.c_int32),
                ('w1', C.c_void_p), ('b1', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p),
                ('w3', C.c_void_p), ('b3', C.c_void_p)]


class PrlPartTables(C.Structure):
    _fields_ = [
        ('n_samples', C.c_int32), ('n_samples_pad', C.c_int32),
        ('sample_xyz', _dp * 3), ('word_bbox', _dp), ('word_valid', _up), ('sample_rank', _ip),
        ('sgrid_origin', C.c_double * 2), ('sgrid_inv_cell', C.c_double),
        ('sgrid_nx', C.c_int32), ('sgrid_ny', C.c_int32), ('sgrid_start', _ip),
        ('n_obs_cells', C.c_int32), ('obs_cell_mask', _up), ('obs_cell_count', _ip),
        ('n_vertices', C.c_int32), ('vertex_xyz', _dp * 3), ('vertex_rank', _ip),
        ('adj_width', C.c_int32), ('vertex_adj', _ip),
        ('vgrid_origin', C.c_double * 2), ('vgrid_inv_cell', C.c_double), ('vgrid_accept', C.c_double),
        ('vgrid_nx', C.c_int32), ('vgrid_ny', C.c_int32), ('vgrid_start', _ip),
        ('n_kd_nodes', C.c_int32), ('kd_node', _ip), ('kd_split', _dp), ('n_kd_points', C.c_int32), ('kd_points', _ip),
        ('kd_box', C.c_double * 6),
        ('n_triangles', C.c_int32), ('tri_records', _dp),
        ('n_collision', C.c_int32), ('n_collision_pad', C.c_int32), ('col_v0e1e2', _dp * 9), ('col_bbox', _fp), ('col_rank', _ip),
        ('col_convex', C.c_int32), ('nbr_width', C.c_int32), ('col_nbr', _ip), ('col_orient', _ip),
        ('n_col_chunks', C.c_int32), ('col_chunk_bbox', _fp),
        ('grid_lo', _dp), ('grid_hi', _dp),
        ('range1', C.c_double * 2), ('range2', C.c_double * 2), ('length_width_ratio', C.c_double),
        ('axis0', C.c_int32), ('axis1', C.c_int32), ('axis2', C.c_int32),
        ('n_start', C.c_int32), ('start_pos', _dp), ('start_quat', _dp),
        ('n_beams', C.c_int32), ('beams', _dp),
    ]


class PrlConfig(C.Structure):
    _fields_ = [
        ('obs_mode', C.c_int32), ('obs_grad', C.c_int32),
        ('action_mode', C.c_int32), ('action_dim', C.c_int32), ('n_discrete', C.c_int32),
        ('termination_mode', C.c_int32), ('turning_penalty', C.c_int32), ('overlap_penalty', C.c_int32),
        ('paint_method', C.c_int32), ('max_episode_len', C.c_int32), ('expected_episode_len', C.c_int32),
        ('auto_reset', C.c_int32), ('color_mode', C.c_int32), ('reserved_', C.c_int32), ('switch_threshold', C.c_double), ('paint_radius', C.c_double),
        ('step_size', C.c_double), ('max_possible_point', C.c_double * 8),
        ('seed', C.c_uint64),
        ('act_delta1', C.c_double * MAX_DISCRETE), ('act_delta2', C.c_double * MAX_DISCRETE),
        ('act_angle', C.c_double * MAX_DISCRETE),
    ]


# every entry point of include/paintrl.h: name -> (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    'prl_abi_version': (C.c_int, []),
    'prl_last_error': (C.c_char_p, []),
    'prl_obs_dim': (C.c_int, [C.POINTER(PrlConfig)]),
    'prl_struct_sizes': (C.c_int, [_ip, _ip]),
    'prl_part_create': (C.c_int, [C.POINTER(PrlPartTables), C.c_int, C.POINTER(_vp)]),
    'prl_part_destroy': (None, [_vp]),
    'prl_part_mask_words': (C.c_int, [_vp]),
    'prl_batch_create': (C.c_int, [C.POINTER(_vp), C.c_int, _ip, C.c_int, C.POINTER(PrlConfig), C.POINTER(_vp)]),
    'prl_batch_destroy': (None, [_vp]),
    'prl_batch_mask_stride': (C.c_int, [_vp]),
    'prl_batch_reset': (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    'prl_batch_step': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'prl_batch_step_occupancy': (C.c_int, [_vp, _ip]),
    'prl_batch_set_pose': (C.c_int, [_vp, C.c_int, _dp, _dp]),
    'prl_batch_observe': (C.c_int, [_vp, _vp, _vp]),
    'prl_policy_act': (C.c_int, [C.POINTER(PrlPolicyWeights), C.c_int, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    'prl_batch_act_step': (C.c_int, [_vp, C.POINTER(PrlPolicyWeights), _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp]),
    'prl_rollout_fragment': (C.c_int, [_vp, C.POINTER(PrlPolicyWeights), C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, C.c_uint64, _vp]),
    'prl_batch_get_mask': (C.c_int, [_vp, _vp, _vp]),
    'prl_batch_get_last_mask': (C.c_int, [_vp, _vp, _vp, _ip, _vp]),
    'prl_batch_get_state': (C.c_int, [_vp, _vp, _vp]),
    'prl_batch_get_thickness': (C.c_int, [_vp, _vp, _vp]),
    'prl_batch_get_returns': (C.c_int, [_vp, _vp, _vp]),
    'prl_ray_batch': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'prl_batch_timing_enable': (C.c_int, [_vp, C.c_int]),
    'prl_batch_timing_read': (C.c_int, [_vp, _dp, C.POINTER(C.c_int64)]),
}
ABI_VERSION = 3          # 3: + prl_batch_step_occupancy, prl_batch_get_last_mask (round 5)
_lib = None


class PaintRLError(RuntimeError):
    pass


def library_path():
    return _build.LIBRARY


def load():
    """Load libpaintrl_hip.so and bind every symbol; raises PaintRLError if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise PaintRLError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(hipcc, gfx950). There is no CPU fallback.' % path)
    lib = C.CDLL(path)
    lax = bool(os.environ.get('PAINTRL_LAX_SYMBOLS'))       # tools/ab_bench.py only: compare against older builds
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if lax:
                continue
            raise PaintRLError('%s does not export %s (stale build?)' % (path, name))
        fn.restype = res
        fn.argtypes = args
    if lib.prl_abi_version() != ABI_VERSION and not lax:
        raise PaintRLError('ABI version mismatch: library %d, binding %d' % (lib.prl_abi_version(), ABI_VERSION))
    cb, tb = C.c_int32(0), C.c_int32(0)
    lib.prl_struct_sizes(C.byref(cb), C.byref(tb))
    if cb.value != C.sizeof(PrlConfig) or tb.value != C.sizeof(PrlPartTables):
        raise PaintRLError('struct layout mismatch: library (%d, %d) vs binding (%d, %d)'
                           % (cb.value, tb.value, C.sizeof(PrlConfig), C.sizeof(PrlPartTables)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise PaintRLError('%s failed (%d): %s' % (what, rc, load().prl_last_error().decode('utf-8', 'replace')))
