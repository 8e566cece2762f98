"""MI355X-native implementation of the CGG (Betrayed by Captions) decoder + mask-prediction hot path.

The directory name carries a hyphen (repo contract); import it as `cgg_amd` (repo-root alias module)
or with importlib.import_module('betrayed-by-captions_amd').

Product code only: HIP kernels (csrc/ -> lib/libcgg_hip.so, C ABI in include/cgg_hip.h), their torch
wrappers (ops.py) and the host-side mirror of the reference's register_module() interface.
The CPU oracle lives in /oracle and is never imported from here.
"""
from . import _lib, ops, runtime  # noqa: F401
from . import registry, config  # noqa: F401
from . import (pixel_decoder, query_decoder, losses, assigner, bert_embeddings,  # noqa: F401
               caption_transformer, mask2former_head, maskformer_fusion_head, backbones, detectors,
               v2l_head, caption_search, pipeline, train, swin)
from .config import Config  # noqa: F401
from .registry import (BACKBONES, BBOX_ASSIGNERS, DETECTORS, HEADS, LOSSES, build_detector,  # noqa: F401
                       build_head, build_loss)

__version__ = '0.1.0'
