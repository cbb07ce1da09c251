"""typlonk_amd -- MI355X (gfx950) backend for TyPLONK's MSM + NTT hot path.

Layout: csrc/ (HIP kernels + C ABI), capi.py (ctypes binding of include/typlonk.h),
dist.py (one-process-per-GPU MSM sharding over torch.distributed/RCCL)."""
from .capi import Context, DeviceBuffer, TyplonkError, load_library, g1_sum_host  # noqa: F401
