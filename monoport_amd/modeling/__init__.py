from .MonoPortNet import MonoPortNet, PIFuNetC, PIFuNetG, QueryBinding, capture_query  # noqa: F401
from .backbones import HGFilter, PIFuHGFilters, PIFuResBlkFilters, ResnetFilter  # noqa: F401
from .geometry import index, orthogonal, perspective  # noqa: F401
from .heads import PIFuNetCMLP, PIFuNetGMLP, SurfaceClassifier  # noqa: F401
from .normalizers import DepthNormalizer, PIFuNomalizer  # noqa: F401
