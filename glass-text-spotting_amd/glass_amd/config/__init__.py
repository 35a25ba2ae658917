from .cfgnode import CfgNode
from .defaults import (get_cfg, add_dataset_config, add_glass_config, add_e2e_config,
                       add_post_process_config, get_glass_cfg)

__all__ = ["CfgNode", "get_cfg", "add_dataset_config", "add_glass_config", "add_e2e_config",
           "add_post_process_config", "get_glass_cfg"]
