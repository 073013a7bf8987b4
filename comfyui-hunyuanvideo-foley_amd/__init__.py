"""MI355X-native HunyuanVideo-Foley sampling path behind the ComfyUI node API of
phazei/ComfyUI-HunyuanVideo-Foley (drop this directory into ComfyUI/custom_nodes/)."""
from .nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
