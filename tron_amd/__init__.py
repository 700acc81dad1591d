"""tron_amd -- MI355X (gfx950) implementation of TRON's 2-D radial gridding / degridding path.

The product is the C-ABI library ``tron_amd/lib/libtronhip.so`` (HIP kernels + host
orchestration + RawArray I/O, built by ``make``) and the ``tron`` command-line driver; this
package is the thin Python mirror used by tests, the bench and scripting:

  tron_amd.lib    ctypes binding of include/tron_hip.h (no CPU fallback)
  tron_amd.ra     RawArray (.ra) files <-> numpy
  tron_amd.shard  slice sharding across ranks / GPUs with a host-side gather
"""
__all__ = ["lib", "ra", "shard"]
