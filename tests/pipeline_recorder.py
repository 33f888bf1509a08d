"""A recording stand-in for the `GlobalSfMpy` module: every attribute access that gets called is logged as
(dotted name of the callee, kinds of the positional arguments).  Objects returned by calls are recorders named after the
call that made them, so `sfm.ReconstructionBuilder(...).CheckView()` logs as "ReconstructionBuilder().CheckView".
Used by tests/golden/make_pipeline_trace.py on the reference's scripts/sfm_pipeline.py (authoring container only) and by
tests/test_pipeline_trace.py on this repo's driver."""
import types


class Recorder:
    def __init__(self, name, log):
        object.__setattr__(self, "_name", name)
        object.__setattr__(self, "_log", log)
        object.__setattr__(self, "_children", {})

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        ch = self._children
        if item not in ch:
            ch[item] = Recorder(self._name + "." + item if self._name else item, self._log)
        return ch[item]

    def __call__(self, *args, **kwargs):
        self._log.append([self._name, [kind_of(a) for a in args], sorted(kwargs)])
        return Recorder(self._name + "()", self._log)

    def __bool__(self):  # `assert estimator.Estimate...()` must hold
        return True

    def __repr__(self):
        return "<recorded %s>" % self._name


def kind_of(a):
    if isinstance(a, Recorder):
        return "obj:" + a._name
    if isinstance(a, bool):
        return "bool"
    if isinstance(a, (int, float, str)):
        return type(a).__name__
    if a is None:
        return "None"
    return "pyobj:" + type(a).__mro__[-2].__name__  # user objects: name of the root base (e.g. a loss class)


def make_module(log, loss_base_name="LossFunction"):
    """A module object whose attributes record; `LossFunction` is a real class so that loss classes can subclass it."""
    mod = types.ModuleType("GlobalSfMpy")
    rec = Recorder("", log)

    class LossFunction:
        def __init__(self):
            pass

    LossFunction.__name__ = loss_base_name
    mod.LossFunction = LossFunction
    mod.tgamma = __import__("math").gamma
    mod.__getattr__ = lambda item: getattr(rec, item)
    return mod
