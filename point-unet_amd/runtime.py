"""Device context: one per process / GPU.  PyTorch supplies device memory and the process group; every
computation goes through the C ABI."""
import ctypes
import threading
import weakref

import numpy as np

from . import _lib

_ctx_lock = threading.Lock()
_contexts = {}


class Context:
    """Owns a ps_context (workspace + stream) on one GPU."""

    def __init__(self, device=0):
        self.device = int(device)
        self._h = ctypes.c_void_p()
        # objects created on this context that own device memory outside torch's allocator (train.Trainer: its activation pool):
        # close() destroys the ones still alive before the context goes
        self._dependents = weakref.WeakSet()
        _lib.check(_lib.lib().ps_create(self.device, ctypes.byref(self._h)))

    def register(self, obj):
        self._dependents.add(obj)

    @property
    def handle(self):
        return self._h

    def use_torch_stream(self):
        import torch
        s = torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib().ps_set_stream(self._h, ctypes.c_void_p(s.cuda_stream)))

    def set_stream(self, stream):
        """Launch on a given torch.cuda.Stream (the context does not own it)."""
        self._stream = stream  # keep it alive
        _lib.check(_lib.lib().ps_set_stream(self._h, ctypes.c_void_p(stream.cuda_stream)))

    def set_deferred_checks(self, on=True):
        """ps_pyramid_build without host synchronisation; status validated by synchronize()."""
        _lib.check(_lib.lib().ps_set_deferred_checks(self._h, 1 if on else 0))

    def set_att_bf16x3(self, on=True):
        """Attentive pooling at d_out = 64 / 128 on bf16 MFMA over exact three-way splits (default) or on the fp32 MFMA."""
        _lib.check(_lib.lib().ps_set_att_bf16x3(self._h, 1 if on else 0))

    def synchronize(self):
        _lib.check(_lib.lib().ps_synchronize(self._h))

    def timing_begin(self, only=None):
        """Arm hipEvent stage timing; `only` restricts it to one stage name."""
        _lib.check(_lib.lib().ps_timing_select(self._h, only.encode() if only else None))
        _lib.check(_lib.lib().ps_timing_begin(self._h))

    def timing_end(self):
        rows = (_lib.PsTimingRow * 128)()
        n = ctypes.c_int(0)
        _lib.check(_lib.lib().ps_timing_end(self._h, rows, 128, ctypes.byref(n)))
        return [(rows[i].name.decode(), rows[i].ms, rows[i].launches) for i in range(n.value)]

    def close(self):
        if self._h:
            for obj in list(self._dependents):
                try:
                    obj.close()
                except Exception:
                    pass
            _lib.lib().ps_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_context(device=None):
    """Process-wide context of `device` (default: torch's current device)."""
    if device is None:
        import torch
        device = torch.cuda.current_device()
    with _ctx_lock:
        if device not in _contexts:
            _contexts[device] = Context(device)
            # torch tensors are allocated, filled and read on torch's current stream: launch there too
            _contexts[device].use_torch_stream()
        return _contexts[device]


def ptr(t):
    """Raw pointer of a torch tensor / numpy array / None as c_void_p."""
    if t is None:
        return ctypes.c_void_p(0)
    if isinstance(t, np.ndarray):
        return ctypes.c_void_p(t.ctypes.data)
    return ctypes.c_void_p(t.data_ptr())
