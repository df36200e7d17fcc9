"""lightx2v_amd: the Wan DiT (and HunyuanVideo-1.5 double-stream) transformer-block forward as gfx950 (MI355X) HIP kernels behind
LightX2V's plugin surface (attention / matmul / norm registries + a WanTransformerInfer-shaped
runner).  Importing the package registers the plugins; the shared library is loaded on first use.
"""

from . import registry_factory  # noqa: F401
from .registry_factory import (  # noqa: F401
    ATTN_WEIGHT_REGISTER,
    LN_WEIGHT_REGISTER,
    MM_WEIGHT_REGISTER,
    RMS_WEIGHT_REGISTER,
    TENSOR_REGISTER,
)
from . import plugins  # noqa: F401,E402
from . import ulysses  # noqa: F401,E402
from .wan_block import (  # noqa: F401,E402
    WanPreInferModuleOutput,
    WanTransformerAttentionBlock,
    WanTransformerInfer,
    WanTransformerWeights,
)
from .hy_block import (  # noqa: F401,E402
    HunyuanVideo15InferModuleOutput,
    HunyuanVideo15TransformerInfer,
    HunyuanVideo15TransformerWeights,
    MMDoubleStreamBlock,
)
from .rope_table import build_rope_table, shard_rope_table  # noqa: F401,E402
from .wan_model import (  # noqa: F401,E402
    MultiDistillModelStruct,
    MultiModelStruct,
    Wan22StepDistillScheduler,
    WanModel,
    WanPostInfer,
    WanPreInfer,
    WanPreWeights,
    WanScheduler,
    WanStepDistillScheduler,
    run_denoise_loop,
)
from .hy_model import (  # noqa: F401,E402
    HunyuanVideo15Model,
    HunyuanVideo15PostInfer,
    HunyuanVideo15PreInfer,
    HunyuanVideo15PreWeights,
    HunyuanVideo15Scheduler,
    HunyuanVideo15StepDistillScheduler,
)

__version__ = "0.1.0"
