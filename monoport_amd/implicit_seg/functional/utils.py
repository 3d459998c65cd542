"""``implicit_seg.functional.utils`` names imported by RTL/main.py:29."""


def plot_mask3D(*args, **kwargs):
    """Debug visualiser in the upstream package (its only call site, RTL/main.py:397-398, is
    commented out).  Out of scope for the reconstruction path."""
    raise NotImplementedError("plot_mask3D is a debugging aid outside the reconstruction path")
