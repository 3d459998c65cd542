"""monoport_amd -- MI355X-native reconstruction hot path of MonoPort (PIFu query over an octree).

Hand-written HIP (gfx950) behind a C-ABI (include/monoport_hip.h); this package is the host-side
mirror of the reference's Python interface for that path:

  monoport_amd.modeling      MonoPortNet / PIFuNetG / PIFuNetC, index, orthogonal, SurfaceClassifier
  monoport_amd.implicit_seg  Seg3dLossless (octree reconstruction engine)
  monoport_amd.recon         pifu_calib, forward_vertices, colorization
  monoport_amd.ops           tensor-level wrappers over the C-ABI
"""
__version__ = "0.1.0"
