"""fast_gicp_amd -- MI355X (gfx950) native VGICP / NDT registration engine.

Drop-in for the device half of koide3/fast_gicp (FastVGICPCudaCore / NDTCudaCore) behind the C ABI in
include/fast_vgicp_hip.h; host-side mirrors of the reference's registration classes live in
fast_gicp_amd.registration (and pygicp).
"""
from . import capi  # noqa: F401
from .capi import NDTCore, VGICPCore  # noqa: F401

__all__ = ["capi", "VGICPCore", "NDTCore"]
