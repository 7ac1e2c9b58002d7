"""The reference's optimiser driver (GSL Fletcher-Reeves CG + its stopping rules) over arbitrary Python functors.

`frcg_minimize` runs libcmaxhip.so's C++ driver (cmx_frcg_minimize: the same loop cmx_frontend_solve /
cmx_backend_solve use) around Python callbacks shaped like gsl_multimin_function_fdf's f / df / fdf.  Used by the
tests to run the identical optimiser over the CPU oracle and over the HIP evaluator, and by hosts that want to plug
a different cost.
"""
import ctypes as C

import numpy as np

from . import _lib

_F = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_void_p)
_DF = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double))
_FDF = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))

FRONTEND = dict(step_size=0.1, tol=0.05, epsabs_grad=1e-3, tolfun=1e-4, max_iterations=50)  # local_optim_contrast_gsl.cpp:106-122
BACKEND = dict(step_size=0.1, tol=0.1, epsabs_grad=1e-4, tolfun=1e-4, max_iterations=50)    # global_optim_contrast_gsl.cpp:41-53


def frcg_minimize(fdf, x0, step_size=0.1, tol=0.05, epsabs_grad=1e-3, tolfun=1e-4, max_iterations=50):
    """fdf(x, want_grad) -> (cost, grad or None), cost being the MINIMISED quantity (-contrast).
    Returns (x, report dict)."""
    x = np.array(x0, dtype=np.float64, order="C", copy=True)
    n = x.size

    def f(xp, _):
        return float(fdf(np.ctypeslib.as_array(xp, (n,)).copy(), False)[0])

    def df(xp, _, gp):
        g = fdf(np.ctypeslib.as_array(xp, (n,)).copy(), True)[1]
        np.ctypeslib.as_array(gp, (n,))[:] = g

    def fdf_c(xp, _, fp, gp):
        c, g = fdf(np.ctypeslib.as_array(xp, (n,)).copy(), True)
        fp[0] = c
        np.ctypeslib.as_array(gp, (n,))[:] = g

    cf, cdf, cfdf = _F(f), _DF(df), _FDF(fdf_c)
    rep = _lib.SolveReport()
    rc = _lib.lib().cmx_frcg_minimize(C.cast(cf, C.c_void_p), C.cast(cdf, C.c_void_p), C.cast(cfdf, C.c_void_p), None, n,
                                      x.ctypes.data_as(C.POINTER(C.c_double)), step_size, tol, epsabs_grad, tolfun,
                                      max_iterations, C.byref(rep))
    _lib.check(None, rc)
    return x, {k: getattr(rep, k) for k, _ in rep._fields_}
