"""ctypes binding of libdfhip.so (C-ABI declared in include/dfhip.h).

This is the only place that touches the shared library.  There is no CPU fallback: if the
library (or a gfx950 device) is missing, loading / creating an engine raises loudly.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdfhip.so')

DFH_OK, DFH_ERR_NOT_PD, DFH_ERR_BAD_ARG, DFH_ERR_HIP, DFH_ERR_JITTER = 0, 1, 2, 3, 4
KERNEL_SE, KERNEL_MATERN, KERNEL_ADDITIVE, KERNEL_PRODUCT, KERNEL_POLY, KERNEL_EXPDECAY = 0, 1, 2, 3, 4, 5
ACQ_MEAN, ACQ_UCB, ACQ_EI, ACQ_PI, ACQ_TTEI, ACQ_STD = 0, 1, 2, 3, 4, 5
GET_L, GET_ALPHA, GET_K = 0, 1, 2
FIT_NO_JITTER, FIT_PROJECT_FIRST, FIT_TRY_BEFORE_PROJECT = 1, 2, 4
LML_X_IS_DEVICE, LML_Y_IS_HOST = 0x100, 0x200       # include/dfhip.h: pointer-kind hints of dfh_gp_lml_batch
T_NAMES = ['kernmat', 'chol', 'solve', 'cross', 'trsm', 'acq', 'ts', 'spare']
INT32_MIN = -2**31
UNIQUE_ID_BYTES = 128

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class KernelDesc(C.Structure):
  """ struct dfh_kernel_desc """
  _fields_ = [('kind', C.c_int32), ('dim', C.c_int32), ('scale', C.c_double), ('nu', C.c_double),
              ('bw', c_double_p), ('n_groups', C.c_int32), ('group_off', c_int32_p),
              ('group_dims', c_int32_p), ('sub_kind', c_int32_p), ('sub_scale', c_double_p),
              ('sub_nu', c_double_p), ('sub_bw', c_double_p), ('group_factor', c_int32_p),
              ('factor_is_sum', c_int32_p), ('factor_scale', c_double_p)]


# name -> (restype, argtypes); must list every symbol include/dfhip.h declares
SIGNATURES = {
  'dfh_abi_version': (C.c_int, []),
  'dfh_device_count': (C.c_int, [C.POINTER(C.c_int)]),
  'dfh_ctx_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
  'dfh_ctx_destroy': (None, [C.c_void_p]),
  'dfh_sync': (C.c_int, [C.c_void_p]),
  'dfh_last_error': (C.c_char_p, []),
  'dfh_device_name': (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
  'dfh_malloc': (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
  'dfh_free': (C.c_int, [C.c_void_p, C.c_void_p]),
  'dfh_memcpy_h2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
  'dfh_memcpy_d2h': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
  'dfh_timer_begin': (C.c_int, [C.c_void_p]),
  'dfh_timer_end': (C.c_int, [C.c_void_p, c_double_p]),
  'dfh_kernel_matrix': (C.c_int, [C.c_void_p, C.POINTER(KernelDesc), C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_int64, C.c_double, C.c_void_p]),
  'dfh_dist_squared': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                 C.c_int64, C.c_void_p]),
  'dfh_gemm': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                         C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p,
                         C.c_int64, C.c_int]),
  'dfh_cholesky': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, c_int64_p]),
  'dfh_stable_cholesky': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, c_int32_p]),
  'dfh_project_psd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p]),
  'dfh_solve_triangular': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                     C.c_int64, C.c_void_p]),
  'dfh_gp_fit': (C.c_int, [C.c_void_p, C.POINTER(KernelDesc), C.c_void_p, C.c_int64, C.c_int64,
                           C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_void_p), c_double_p,
                           c_int32_p]),
  'dfh_gp_lml_batch': (C.c_int, [C.c_void_p, C.POINTER(KernelDesc), C.c_int32, C.c_void_p, C.c_int64,
                                 C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p]),
  'dfh_gp_append': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                              C.POINTER(C.c_void_p), c_double_p, c_int32_p]),
  'dfh_mem_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
  'dfh_gp_fit_gram': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_int,
                                C.POINTER(C.c_void_p), c_double_p, c_int32_p]),
  'dfh_gp_predict_gram': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
  'dfh_gp_predict_covar_gram': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
  'dfh_gp_add_ucb_all': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
  'dfh_rand_mt19937_uniform': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_int64, C.c_void_p, C.c_void_p]),
  'dfh_mt19937_advance': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
  'dfh_rand_mt19937_normal': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
  'dfh_rand_philox_uniform': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
  'dfh_gp_free': (C.c_int, [C.c_void_p]),
  'dfh_gp_get': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
  'dfh_gp_n': (C.c_int64, [C.c_void_p]),
  'dfh_gp_refine_steps': (C.c_int, [C.c_void_p, c_int32_p]),
  'dfh_gp_predict': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p]),
  'dfh_gp_predict_covar': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_void_p]),
  'dfh_gp_acq_argmax': (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p,
                                  c_double_p, c_int64_p]),
  'dfh_gp_ts': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_double,
                          C.c_void_p, C.c_void_p, c_double_p, c_int64_p, c_int32_p]),
  'dfh_gp_add_ucb_group': (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int64,
                                     C.c_void_p, c_double_p, c_int64_p]),
  # multi-GPU: host-only contract, one-process-per-GPU communicator, in-process fan-out
  'dfh_shard_bounds': (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int64, c_int64_p, c_int64_p]),
  'dfh_reduce_argmax': (C.c_int, [c_double_p, c_int64_p, C.c_int, c_double_p, c_int64_p]),
  'dfh_comm_unique_id': (C.c_int, [C.c_void_p]),
  'dfh_comm_create': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
  'dfh_comm_destroy': (None, [C.c_void_p]),
  'dfh_comm_rank': (C.c_int, [C.c_void_p]),
  'dfh_comm_size': (C.c_int, [C.c_void_p]),
  'dfh_comm_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
  'dfh_comm_allgather_argmax': (C.c_int, [C.c_void_p, C.c_double, C.c_int64, c_double_p, c_int64_p]),
  'dfh_comm_allgather_f64': (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p]),
  'dfh_comm_allreduce_max': (C.c_int, [C.c_void_p, c_double_p, C.c_int]),
  'dfh_comm_barrier': (C.c_int, [C.c_void_p]),
  'dfh_mgpu_create': (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
  'dfh_mgpu_destroy': (None, [C.c_void_p]),
  'dfh_mgpu_size': (C.c_int, [C.c_void_p]),
  'dfh_mgpu_ctx': (C.c_void_p, [C.c_void_p, C.c_int]),
  'dfh_mgpu_gp': (C.c_void_p, [C.c_void_p, C.c_int]),
  'dfh_mgpu_comm': (C.c_void_p, [C.c_void_p, C.c_int]),
  'dfh_mgpu_sync': (C.c_int, [C.c_void_p]),
  'dfh_mgpu_fit': (C.c_int, [C.c_void_p, C.POINTER(KernelDesc), C.POINTER(C.c_void_p), C.c_int64, C.c_int64,
                             C.POINTER(C.c_void_p), C.c_double, C.c_int, c_double_p, c_int32_p]),
  'dfh_mgpu_free_fit': (C.c_int, [C.c_void_p]),
  'dfh_mgpu_ts': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_int64_p, C.c_int64, C.POINTER(C.c_void_p),
                            C.c_double, c_double_p, c_int64_p, c_double_p, c_int64_p]),
  'dfh_mgpu_acq_argmax': (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.POINTER(C.c_void_p), c_int64_p,
                                    C.c_double, c_double_p, c_int64_p, c_double_p, c_int64_p]),
  'dfh_mgpu_allgather_argmax': (C.c_int, [C.c_void_p, c_double_p, c_int64_p, c_double_p, c_int64_p]),
  'dfh_ctx_timings': (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
  'dfh_ctx_counters': (C.c_int, [C.c_void_p, c_int64_p]),
  'dfh_ctx_gemm_profile': (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
}

_lib = None


class DfhipError(RuntimeError):
  """ HIP / library failure (DFH_ERR_HIP). """


def load():
  """ Loads libdfhip.so (once). Raises if it has not been built: there is no fallback path. """
  global _lib
  if _lib is not None:
    return _lib
  # DFH_LIB: another build of the same sources (the diagnostics build libdfhip_dbg.so of kernel work)
  path = os.environ.get('DFH_LIB') or LIB_PATH
  if path != LIB_PATH:
    sys.stderr.write('dragonfly_amd: DFH_LIB is set -- loading %s instead of the packaged %s\n' % (path, LIB_PATH))
  if not os.path.exists(path):
    raise ImportError('%s not found. Build it with `python -m dragonfly_amd.build` '
                      '(hipcc, gfx950). dragonfly_amd has no CPU fallback.' % path)
  lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
  _lib = lib
  return lib


def last_error():
  msg = load().dfh_last_error()
  return msg.decode('utf-8', 'replace') if msg else ''


def check(rc):
  """ Maps a status code to the exception the reference would raise at that point. """
  if rc == DFH_OK:
    return
  msg = last_error()
  if rc == DFH_ERR_NOT_PD:
    raise np.linalg.LinAlgError(msg or 'Matrix is not positive definite')
  if rc in (DFH_ERR_BAD_ARG, DFH_ERR_JITTER):
    raise ValueError(msg)
  raise DfhipError(msg or 'libdfhip error %d' % rc)


def device_count():
  n = C.c_int(0)
  load().dfh_device_count(C.byref(n))
  return n.value
