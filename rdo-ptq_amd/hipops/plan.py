"""Recorded op list executed natively (rdo_plan_* of include/rdo_ptq_hip.h): one call enqueues whole calibration
iterations, optionally through a captured hipGraph -- no per-kernel Python or ctypes work in the hot loop."""
import contextlib
import ctypes as C

import torch

from . import _lib as L


class Plan:
    def __init__(self):
        self._h = C.c_void_p(L.lib().rdo_plan_create())
        self._keep = []          # tensors referenced by recorded ops must outlive the plan

    def keep(self, *tensors):
        self._keep.extend(tensors)

    @contextlib.contextmanager
    def record(self):
        L.check(L.lib().rdo_plan_begin_record(self._h), "rdo_plan_begin_record")
        try:
            yield self
        finally:
            L.check(L.lib().rdo_plan_end_record(self._h), "rdo_plan_end_record")

    @staticmethod
    @contextlib.contextmanager
    def eager():
        """Inside a `record()` scope: the calls made in this block are launched now, not recorded (no-op outside a recording)."""
        was = int(L.lib().rdo_plan_suspend_record(1))
        try:
            yield
        finally:
            if not was:
                L.lib().rdo_plan_suspend_record(0)

    @property
    def num_ops(self):
        return int(L.lib().rdo_plan_num_ops(self._h))

    def run(self, n_iters=1, graph=True):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib().rdo_plan_run(self._h, int(n_iters), int(bool(graph)), s), "rdo_plan_run")

    def prepare(self, n_iters):
        """Build the graph(s) `run(n_iters)` will replay now, launching nothing (set-up work: a timed loop should not pay for it)."""
        L.check(L.lib().rdo_plan_prepare(self._h, int(n_iters)), "rdo_plan_prepare")

    def run_then(self, other, graph=True):
        """One iteration of this plan followed by one of `other` in one graph launch."""
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib().rdo_plan_run_then(self._h, other._h, int(bool(graph)), s), "rdo_plan_run_then")

    def op_info(self):
        """[(tag, flops, bytes)] of the recorded ops."""
        out = []
        for i in range(self.num_ops):
            tag, fl, by = C.c_char_p(), C.c_double(), C.c_double()
            L.check(L.lib().rdo_plan_op_info(self._h, i, C.byref(tag), C.byref(fl), C.byref(by)), "rdo_plan_op_info")
            out.append((tag.value.decode(), fl.value, by.value))
        return out

    def profile(self):
        """Run ONE eager iteration with a hipEvent pair around every op -> per-op milliseconds (synchronises)."""
        ms = (C.c_float * self.num_ops)()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib().rdo_plan_profile(self._h, ms, s), "rdo_plan_profile")
        return list(ms)

    def __del__(self):
        try:
            if self._h:
                L.lib().rdo_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass
