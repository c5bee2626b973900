"""Host mirrors of the one-element switch buffers (`fake_quant_enabled`, `observer_enabled`, SmoothQuant's
`enabled` / `dynamic` / `fused_to_weight`).

The reference (and torch.ao's FakeQuantize it derives from) keeps these switches as registered buffers and reads
them in every forward (`if self.fake_quant_enabled[0] == 1`, numerical/cast.py:268-277).  Once the module lives
on the GPU each such read copies one byte back to the host and WAITS for the stream — two or three full
pipeline drains per cast.  Here the buffers stay (same state_dict keys), but every write goes through
`_set_flag`, which also records the value in the instance `__dict__`; forward paths read only the mirror.
Code that pokes a buffer directly must call `refresh_flags()` afterwards; `load_state_dict` does it by itself.
"""


class FastAttr:
    """nn.Module.__getattr__ -- what every `self.weight_cast`, `self.smoothquant`, `self.scale` costs, because submodules, parameters and
    buffers live in three dicts outside `__dict__` -- restated for the lookups this package makes ~260 times per forward of a decoder
    layer (round 6: 70 of 1030 profiled microseconds, profiles/r06_eager_profile_opt125m.txt): the three dicts straight from
    `__dict__`, submodules first (the commonest miss), the same AttributeError at the end.  Same lookup rule, nothing cached."""

    def __getattr__(self, name):
        d = self.__dict__
        try:
            m = d["_modules"]
            if name in m:
                return m[name]
            m = d["_parameters"]
            if name in m:
                return m[name]
            m = d["_buffers"]
            if name in m:
                return m[name]
        except KeyError:   # (before nn.Module.__init__ ran)
            pass
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")


class HostFlags(FastAttr):
    _flag_names = ()
    #: zero-dim float buffers mirrored the same way (SmoothQuant's `migration_strength`, `scale_min`: read with `float(...)` in every
    #: calibration / dynamic forward -- a device->host copy and a stream drain each once the module is on the GPU, and not capturable)
    _scalar_names = ()

    def _set_scalar(self, name: str, value: float) -> None:
        buf = getattr(self, name)
        buf.fill_(float(value))
        # the mirror holds what the BUFFER holds (0.3 -> 0.30000001192 in a float32 buffer), whichever path wrote it last;
        # rounded on the host: no device read
        import torch
        self.__dict__["_h_" + name] = float(torch.tensor(float(value), dtype=buf.dtype))

    def _scalar(self, name: str) -> float:
        try:
            return self.__dict__["_h_" + name]
        except KeyError:   # a module whose __dict__ predates the mirrors (whole-module pickle of an earlier version), or a buffer
            self.refresh_flags()   # that was assigned directly: read the buffers once
            return self.__dict__["_h_" + name]

    def _set_flag(self, name: str, value) -> None:
        getattr(self, name)[0] = 1 if value else 0
        self.__dict__["_h_" + name] = 1 if value else 0

    def _flag(self, name: str) -> int:
        try:
            return self.__dict__["_h_" + name]
        except KeyError:
            self.refresh_flags()
            return self.__dict__["_h_" + name]

    def refresh_flags(self) -> None:
        for n in self._flag_names:
            self.__dict__["_h_" + n] = int(getattr(self, n)[0])
        for n in self._scalar_names:
            self.__dict__["_h_" + n] = float(getattr(self, n))

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.refresh_flags()
