from .default import _C as cfg
from .default import update_config, hrnet_extra
from .node import CfgNode
