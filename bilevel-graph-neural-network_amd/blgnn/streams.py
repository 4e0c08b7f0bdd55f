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
"""

import atexit
import ctypes

import torch

__all__ = ["cu_partition_streams"]

_XCDS, _CUS_PER_XCD, _GRANULE = 8, 32, 4
_LIVE = []          # (device, raw handle) of every masked stream: destroyed at interpreter exit


def _destroy_all():
    """Masked streams are raw HIP streams torch does not own; leaving them to process teardown
    crashes tools that finalise after the runtime (rocprofv3 segfaults in __cxa_finalize)."""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
        torch.cuda.synchronize()
        while _LIVE:
            device, handle = _LIVE.pop()
            with torch.cuda.device(device):
                hip.hipStreamDestroy(handle)
    except Exception:                                     # exit path: never raise
        pass


atexit.register(_destroy_all)


def _masked_stream(device, cu_lo, cu_hi):
    bits = 0
    for i in range(_XCDS * _CUS_PER_XCD):
        if cu_lo <= i // _XCDS < cu_hi:
            bits |= 1 << i
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * w)) & 0xffffffff for w in range(8)])
    hip = ctypes.CDLL("libamdhip64.so")                 # the runtime torch already loaded
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 8, words)
    if rc != 0 or not handle.value:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    _LIVE.append((device, handle.value))
    return torch.cuda.ExternalStream(handle.value, device=device)


def cu_partition_streams(device, first_cus_per_xcd):
    """-> (stream_a, stream_b): stream_a owns CUs [0, first_cus_per_xcd) of every XCD, stream_b the rest.
    ``first_cus_per_xcd`` is rounded to the hardware's granule of 4 and kept inside [4, 28].  Returns
    None when the device is not an 8 x 32-CU part (the mask layout above is specific to it) or the HIP
    runtime does not offer masked streams; the caller then keeps ordinary streams."""
    props = torch.cuda.get_device_properties(device)
    if props.multi_processor_count != _XCDS * _CUS_PER_XCD:
        return None
    n = int(round(first_cus_per_xcd / _GRANULE)) * _GRANULE
    n = min(max(n, _GRANULE), _CUS_PER_XCD - _GRANULE)
    try:
        return _masked_stream(device, 0, n), _masked_stream(device, n, _CUS_PER_XCD)
    except (OSError, AttributeError, RuntimeError):      # runtime without the extension: ordinary streams
        return None
