from .MVGenModel import MultiViewBaseModel  # noqa: F401
from .modules import WarpAttn  # noqa: F401
from .utils import get_coords, get_masks  # noqa: F401
