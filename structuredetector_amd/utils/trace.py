"""Profiler ranges around the stages of the hot path (the reference has no tracing; `rocprofv3 --marker-trace`
needs roctx ranges to make a step navigable).  Thin mirror of the C ABI's `sd_range_push / sd_range_pop` (csrc/sd_trace.cpp), which
resolve the marker library at run time and only when the environment holds SDNET_ROCTX=1.  Disabled (the default), `span()` costs one
attribute read and returns a shared no-op context manager: nothing is pushed, no ctypes call is made.

    SDNET_ROCTX=1 rocprofv3 --marker-trace --kernel-trace -d out -- python3 bench.py --steps 3 --warmup 1
"""
from .. import _lib as L

_state = {"on": None}


class _Null:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _Span:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        L.lib().sd_range_push(self.name)
        return self

    def __exit__(self, *exc):
        L.lib().sd_range_pop()
        return False


_NULL = _Null()


def enabled():
    on = _state["on"]
    if on is None:
        on = _state["on"] = bool(L.lib().sd_range_enabled())
    return on


def span(name):
    """Context manager: a roctx range `name` on the calling thread while ranges are enabled, a no-op otherwise."""
    if _state["on"] is False:
        return _NULL
    return _Span(name.encode()) if enabled() else _NULL


def push(name):
    if enabled():
        L.lib().sd_range_push(name.encode())


def pop():
    if enabled():
        L.lib().sd_range_pop()


def mark(name):
    if enabled():
        L.lib().sd_range_mark(name.encode())
