"""rtm3d_amd - MI355X (gfx950) native RTM3D inference hot path behind the call surface of
hitfeelee/rtm3d: create_model(cfg), Model.forward / inference, optim_decode_bbox3d, ParamList,
CheckPointer.  All arithmetic runs in hand-written HIP kernels (rtm3d_amd/csrc, C ABI in
include/rtm3d_hip.h); Python orchestrates."""
from .config import CONFIGS, CfgNode, kitti_config           # noqa: F401
from .model_factory import create_model                       # noqa: F401
from .model import Model                                      # noqa: F401
from .ParamList import ParamList                              # noqa: F401
from . import model_utils, weights, check_point               # noqa: F401

__version__ = '0.1.0'
