"""models.<NAME>.get_pose_net(cfg, is_train) - same selection contract as reference tools/train.py:92-94."""
from . import pose_resnet
from . import pose_hrnet
from . import pose_hrnet_coam
from . import transpose_h
