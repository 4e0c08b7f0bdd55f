"""HIP streams restricted to a slice of every XCD's compute units.

The two halves of a step (molecular sweep / interaction level + pair scorer) run on two streams.
Both are made of kernels that fill the chip with latency-bound workgroups, so on ordinary streams
they mostly take turns.  Giving each stream its own compute units lets them really run side by
side: the fused GIN layer needs two rounds of workgroups on 256 CUs and still two on 192, while the
small interaction-level kernels stop queueing behind 1 138 GIN workgroups.

MI355X: 8 XCDs x 32 CUs.  In ``hipExtStreamCreateWithCUMask`` bit ``i`` of the mask addresses CU
``i // 8`` of XCD ``i % 8`` (measured: a mask that leaves an XCD without CUs is ignored for that XCD,
and CUs are granted in groups of four per XCD), and workgroups are dealt round-robin to the XCDs, so
a partition has to take the same CUs on every XCD or the poorest XCD sets the pace.

Ownership: a ``MaskedStreamPair`` owns its two raw HIP streams (created and destroyed through
libblgnn, i.e. the HIP runtime the library is linked against) and ``close()`` destroys them after a
device synchronise; ``BiLevelEngine.close()`` calls it.  Raw streams that outlive their owner crash
tools that finalise after the runtime (rocprofv3 segfaulted in ``__cxa_finalize``), so every live
pair is also closed at interpreter exit as a last resort.
"""

import atexit
import ctypes
import weakref

import torch

from . import _lib

__all__ = ["cu_partition_streams", "MaskedStreamPair"]

_XCDS, _CUS_PER_XCD, _GRANULE = 8, 32, 4
_LIVE = weakref.WeakSet()


def _create(device, cu_lo, cu_hi):
    bits = 0
    for i in range(_XCDS * _CUS_PER_XCD):
        if cu_lo <= i // _XCDS < cu_hi:
            bits |= 1 << i
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * w)) & 0xffffffff for w in range(8)])
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        _lib.call("blgnn_stream_create_cu_mask", words, 8, ctypes.byref(handle))
    return handle.value


class MaskedStreamPair(object):
    """(first, second): ``first`` owns CUs [0, n) of every XCD, ``second`` the rest."""

    def __init__(self, device, n):
        self.device, self.cus = torch.device(device), n
        self._handles = []
        try:
            self._handles.append(_create(self.device, 0, n))
            self._handles.append(_create(self.device, n, _CUS_PER_XCD))
        except Exception:
            self.close()
            raise
        self.first = torch.cuda.ExternalStream(self._handles[0], device=self.device)
        self.second = torch.cuda.ExternalStream(self._handles[1], device=self.device)
        _LIVE.add(self)

    def close(self):
        """Synchronise the device and destroy both streams; idempotent."""
        handles, self._handles = self._handles, []
        if not handles:
            return
        self.first = self.second = None
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            for h in handles:
                _lib.call("blgnn_stream_destroy", h)


def _close_all():
    for pair in list(_LIVE):                              # exit path: never raise
        try:
            pair.close()
        except Exception:
            pass


atexit.register(_close_all)


def cu_partition_streams(device, first_cus_per_xcd):
    """-> MaskedStreamPair, or None when the device is not an 8 x 32-CU part (the mask layout above is
    specific to it) or the runtime refuses masked streams; the caller then keeps ordinary streams.
    ``first_cus_per_xcd`` is rounded to the hardware's granule of 4 and kept inside [4, 28]."""
    props = torch.cuda.get_device_properties(device)
    if props.multi_processor_count != _XCDS * _CUS_PER_XCD:
        return None
    n = int(round(first_cus_per_xcd / _GRANULE)) * _GRANULE
    n = min(max(n, _GRANULE), _CUS_PER_XCD - _GRANULE)
    try:
        return MaskedStreamPair(device, n)
    except (_lib.BlgnnError, ValueError):
        return None
