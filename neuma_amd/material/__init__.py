from .meta import InvariantFullMetaElasticity, InvariantFullMetaPlasticity, MLPBlock, MaterialFunction
from .preset import ComposeMaterial
from .loralib import LinearLoRA, mark_only_lora_as_trainable, lora_state_dict, replace_with_linear_lora, init_linear_lora
