"""Robust losses with the class names and constructor signatures of the reference's
scripts/loss_functions.py (a13 in SURVEY.md section 8), written for this build.

Every class still answers ``Evaluate(sq_norm, out)`` in Python (out[0..2] = rho, rho', rho'', Ceres
convention: the cost of a residual block is rho(s)/2, reference loss_functions.py:8-31) AND exposes
``native_program()``: the descriptor (include/gsfm_rot.h, gsfm_loss_node) that the HIP kernels
evaluate per edge on the device, so no per-edge Python callback is needed.  A user subclass that
only defines ``Evaluate`` still works: it is evaluated on the host per edge, like the reference's
pybind trampoline does (bind_src/GlobalSfMpy.cpp:33-65).

Usage is the reference's:  ``from loss_functions import *``;  ``MAGSACWeightBasedLoss(0.02)``.
"""
import math
import sys

import os


def _compiled_module():
    """The compiled drop-in module `GlobalSfMpy` (top-level name, as scripts/sfm_pipeline.py:5-6 imports it).  It is
    looked up on sys.path first and then next to this file, so that the loss classes derive from its LossFunction no
    matter which module was imported first."""
    try:
        import GlobalSfMpy
        return GlobalSfMpy
    except ImportError:
        pass
    here = os.path.dirname(os.path.abspath(__file__))
    if any(f.startswith("_GlobalSfMpy") and f.endswith(".so") for f in os.listdir(here)):
        sys.path.append(here)
        try:
            import GlobalSfMpy
            return GlobalSfMpy
        except ImportError:
            pass
    return None


sfm = _compiled_module()
if sfm is not None:
    _Base = sfm.LossFunction
else:  # pure-Python use (module not built): same protocol, no compiled base needed
    class _Base(object):
        def __init__(self):
            pass

from globalsfmpy_amd import _abi as _k

_TINY = sys.float_info.min

__all__ = ["TrivialLoss", "HuberLoss", "SoftLOneLoss", "CauchyLoss", "ArctanLoss", "TolerantLoss", "TukeyLoss",
           "LOneHalfLoss", "LTwoLoss", "GemanMcClureLoss", "ComposedLoss", "ScaledLoss",
           "MAGSACWeightBasedLoss", "MAGSACWeightBasedLoss4", "MAGSACWeightBasedLoss9"]


class _Loss(_Base):
    """Common plumbing: subclasses give _rho(s) -> (rho, rho', rho'') and _node()."""

    def __init__(self):
        _Base.__init__(self)

    def Evaluate(self, sq_norm, out):
        r = self._rho(sq_norm)
        out[0] = r[0]
        out[1] = r[1]
        out[2] = r[2]

    def native_program(self):
        if _user_override_without_descriptor(self):
            return None
        return [self._node()]


def _user_override_without_descriptor(loss):
    """True when a class derived from one of the shipped losses OUTSIDE this module redefines the function (Evaluate or _rho) without
    also redefining its device descriptor: the device must then not run the parent's program -- the reference always calls the
    Python Evaluate through its trampoline (bind_src/GlobalSfMpy.cpp:33-65), and so does the host-callback path here."""
    for klass in type(loss).__mro__:
        if klass.__module__ == __name__:
            return False                      # reached the library's own class: nothing above it changed the function
        d = klass.__dict__
        if ("Evaluate" in d or "_rho" in d) and not ("native_program" in d or "_node" in d):
            return True
    return False


class TrivialLoss(_Loss):
    """rho(s) = s  (reference :47)."""

    def _rho(self, s):
        return (s, 1.0, 0.0)

    def _node(self):
        return (_k.LOSS_TRIVIAL, 0.0, 0.0, 0.0)


class HuberLoss(_Loss):
    """rho = s inside s <= a^2, 2a sqrt(s) - a^2 outside  (reference :56)."""

    def __init__(self, a):
        self.a = a
        self.b = a * a
        _Loss.__init__(self)

    def _rho(self, s):
        if s > self.b:
            root = math.sqrt(s)
            d1 = max(self.a / root, _TINY)
            return (2 * self.a * root - self.b, d1, -d1 / (2.0 * s))
        return (s, 1.0, 0.0)

    def _node(self):
        return (_k.LOSS_HUBER, self.a, 0.0, 0.0)


class SoftLOneLoss(_Loss):
    """rho = 2 a^2 (sqrt(1 + s/a^2) - 1)  (reference :74)."""

    def __init__(self, a):
        self.a = a
        self.b = a * a
        self.c = 1.0 / self.b
        _Loss.__init__(self)

    def _rho(self, s):
        t = 1.0 + s * self.c
        root = math.sqrt(t)
        d1 = max(1.0 / root, _TINY)
        return (2.0 * self.b * (root - 1.0), d1, -(self.c * d1) / (2.0 * t))

    def _node(self):
        return (_k.LOSS_SOFT_L1, self.a, 0.0, 0.0)


class CauchyLoss(_Loss):
    """rho = a^2 log(1 + s/a^2)  (reference :88)."""

    def __init__(self, a):
        self.a = a
        self.b = a * a
        self.c = 1.0 / self.b
        _Loss.__init__(self)

    def _rho(self, s):
        t = 1.0 + s * self.c
        inv = 1.0 / t
        return (self.b * math.log(t), max(inv, _TINY), -self.c * (inv * inv))

    def _node(self):
        return (_k.LOSS_CAUCHY, self.a, 0.0, 0.0)


class ArctanLoss(_Loss):
    """rho = a atan2(s, a)  (reference :101)."""

    def __init__(self, a):
        self.a = a
        self.b = 1 / (a * a)
        _Loss.__init__(self)

    def _rho(self, s):
        t = 1 + s * s * self.b
        inv = 1.0 / t
        return (self.a * math.atan2(s, self.a), max(inv, _TINY), -2.0 * s * self.b * (inv * inv))

    def _node(self):
        return (_k.LOSS_ARCTAN, self.a, 0.0, 0.0)


class TolerantLoss(_Loss):
    """rho = b log(1 + exp((s - a)/b)) - b log(1 + exp(-a/b)), linear beyond (s-a)/b > 36.7  (reference :114)."""

    def __init__(self, a, b):
        assert a >= 0
        assert b > 0
        self.a = a
        self.b = b
        self.c = b * math.log(1 + math.exp(-a / b))
        _Loss.__init__(self)

    def _rho(self, s):
        x = (s - self.a) / self.b
        if x > 36.7:
            return (s - self.a - self.c, 1.0, 0.0)
        e_x = math.exp(x)
        return (self.b * math.log(1.0 + e_x) - self.c, max(e_x / (1.0 + e_x), _TINY),
                0.5 / (self.b * (1.0 + math.cosh(x))))

    def _node(self):
        return (_k.LOSS_TOLERANT, self.a, self.b, 0.0)


class TukeyLoss(_Loss):
    """rho = a^2/6 (1 - (1 - s/a^2)^3) for s <= a^2, a^2/6 beyond  (reference :167)."""

    def __init__(self, a):
        self.a = a
        self.a_squared = a * a
        _Loss.__init__(self)

    def _rho(self, s):
        if s <= self.a_squared:
            v = 1.0 - s / self.a_squared
            v2 = v * v
            return (self.a_squared / 6.0 * (1.0 - v2 * v), 0.5 * v2, -1.0 / self.a_squared * v)
        return (self.a_squared / 6.0, 0.0, 0.0)

    def _node(self):
        return (_k.LOSS_TUKEY, self.a, 0.0, 0.0)


class LOneHalfLoss(_Loss):
    """rho = 2 a sqrt(a) s^(1/4); derivatives evaluated at max(s, 0.01)  (reference :187)."""

    def __init__(self, a):
        self.a = a
        self.sqrt_a = math.sqrt(a)
        _Loss.__init__(self)

    def _rho(self, s):
        r0 = 2.0 * self.a * self.sqrt_a * pow(s, 0.25)
        if s < 0.01:
            s = 0.01
        return (r0, 0.5 * pow(self.a, -1.5) * pow(s, -0.75), -0.375 * self.a * self.sqrt_a * pow(s, -1.75))

    def _node(self):
        return (_k.LOSS_LONE_HALF, self.a, 0.0, 0.0)


class LTwoLoss(_Loss):
    """rho = s^2 / (2 a^2); sigma2 is accepted and ignored, as in the reference (:216-237)."""

    def __init__(self, a, sigma2):
        self.a = a
        self.a_sq = a * a
        _Loss.__init__(self)

    def _rho(self, s):
        return (s * s / (self.a_sq * 2.0), s / self.a_sq, 1 / self.a_sq)

    def _node(self):
        return (_k.LOSS_LTWO, self.a, 0.0, 0.0)


class GemanMcClureLoss(_Loss):
    """rho = a^2 sigma2 s / (2 (s + a^2 sigma2))  (reference :239)."""

    def __init__(self, a, sigma2):
        self.a = a
        self.a_sq = a * a
        self.sigma2 = sigma2
        _Loss.__init__(self)

    def _rho(self, s):
        t = s / self.a_sq + self.sigma2
        return (self.a_sq * self.sigma2 * s / (2.0 * (s + self.a_sq * self.sigma2)),
                (self.sigma2 ** 2) / (2.0 * t ** 2),
                -(self.sigma2 ** 2) / (self.a_sq * t ** 3))

    def _node(self):
        return (_k.LOSS_GEMAN_MCCLURE, self.a, self.sigma2, 0.0)


def _program_of(loss):
    prog = loss.native_program() if hasattr(loss, "native_program") else None
    return prog


class ComposedLoss(_Loss):
    """rho(s) = f(g(s))  (reference :250)."""

    def __init__(self, f, g):
        self.f = f
        self.g = g
        _Loss.__init__(self)

    def _rho(self, s):
        og = [0.0, 0.0, 0.0]
        of = [0.0, 0.0, 0.0]
        self.g.Evaluate(s, og)
        self.f.Evaluate(og[0], of)
        return (of[0], of[1] * og[1], of[2] * og[1] * og[1] + of[1] * og[2])

    def native_program(self):
        if _user_override_without_descriptor(self):
            return None
        pf, pg = _program_of(self.f), _program_of(self.g)
        if pf is None or pg is None:
            return None
        return pg + [(_k.LOSS_OP_PUSH_ARG, 0.0, 0.0, 0.0)] + pf + [(_k.LOSS_OP_COMPOSE, 0.0, 0.0, 0.0)]


class ScaledLoss(_Loss):
    """rho(s) = a * inner(s)  (reference :267)."""

    def __init__(self, rho, a):
        self.rho = rho
        self.a = a
        _Loss.__init__(self)

    def _rho(self, s):
        o = [0.0, 0.0, 0.0]
        self.rho.Evaluate(s, o)
        return (o[0] * self.a, o[1] * self.a, o[2] * self.a)

    def native_program(self):
        if _user_override_without_descriptor(self):
            return None
        p = _program_of(self.rho)
        if p is None:
            return None
        return p + [(_k.LOSS_OP_SCALE, self.a, 0.0, 0.0)]


# ---------------------------------------------------------------------------------------------
# MAGSAC sigma-consensus losses (reference :285, :344, :402).  One implementation, three degrees
# of freedom; the tables Gamma((nu-1)/2, x/1000) come from the library (regenerated analytically,
# include/gsfm_rot.h gsfm_magsac_table) or, when the compiled module is present, from its attrs.
# ---------------------------------------------------------------------------------------------
_tables = {}


def _magsac_data(nu):
    if nu not in _tables:
        from globalsfmpy_amd import solver as _solver
        C, q, gk = _solver.magsac_constants(nu)
        _tables[nu] = (C, q, gk, _solver.magsac_table(nu))
    return _tables[nu]


class _MagsacLoss(_Loss):
    _nu = 3

    def _setup(self, sigma, inverse):
        nu = float(self._nu)
        C, q, gk, table = _magsac_data(self._nu)
        self.sigma_max = sigma
        self.nu = nu
        self.use_weight_inverse = inverse
        self._q, self._gk, self._table = q, gk, table
        self.squared_sigma = sigma * sigma
        self.squared_sigma_max_2 = 2.0 * self.squared_sigma
        self.cubed_sigma_max = self.squared_sigma * sigma
        self.dof_minus_one_per_two = (nu - 1.0) / 2.0
        self.C_times_two_ad_dof = C * (2 ** self.dof_minus_one_per_two)
        self.one_over_sigma = self.C_times_two_ad_dof / sigma
        self.gamma_value = math.gamma(self.dof_minus_one_per_two)
        self.gamma_difference = self.gamma_value - gk
        self.weight_zero = self.one_over_sigma * self.gamma_difference
        _Loss.__init__(self)

    def _rho(self, s_in):
        cut = self._q * self._q * self.squared_sigma
        flat = s_in > cut
        if flat:
            s_in = cut
        x = round(1000.0 * s_in / self.squared_sigma_max_2)  # half-to-even, like the reference's Python
        x = min(x, len(self._table))
        s = x * self.squared_sigma_max_2 / 1000.0
        u = s / self.squared_sigma_max_2
        w = self.one_over_sigma * (self._table[x] - self._gk)
        w1 = -self.C_times_two_ad_dof * (u ** (self.nu / 2 - 1.5)) * math.exp(-u) / (2 * self.cubed_sigma_max)
        if s < 1e-7:
            s = 1e-7
        u = s / self.squared_sigma_max_2
        w2 = 2.0 * self.C_times_two_ad_dof * (u ** (self.nu / 2 - 1.5)) * \
            (1.0 / self.squared_sigma - (self.nu - 3) / s) * math.exp(-u) / (8 * self.cubed_sigma_max)
        if self.use_weight_inverse:
            r = [1.0 / w, -1.0 / (w * w) * w1, 2.0 / (w * w * w) * w1 * w1 - w2 / (w * w)]
        else:
            r = [self.weight_zero - w, -w1, -w2]
            if r[1] == 0:
                r[1] = 0.00001
        if flat:
            r[1] = 0.00001
            r[2] = 0.0
        return tuple(r)

    def _node(self):
        return (_k.LOSS_MAGSAC, self.sigma_max, float(self._nu), 1.0 if self.use_weight_inverse else 0.0)


class MAGSACWeightBasedLoss(_MagsacLoss):
    _nu = 3

    def __init__(self, sigma, inverse=False):
        self._setup(sigma, inverse)


class MAGSACWeightBasedLoss4(_MagsacLoss):
    _nu = 4

    def __init__(self, sigma, inverse=True):  # the reference's default for nu=4 is True (:345)
        self._setup(sigma, inverse)


class MAGSACWeightBasedLoss9(_MagsacLoss):
    _nu = 9

    def __init__(self, sigma, inverse=False):
        self._setup(sigma, inverse)
