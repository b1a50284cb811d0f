"""vit-tf hot path, MI355X-native: DINO ViT K-feature volumes + similarity queries through hand-written
HIP kernels (libvittf.so, C ABI in include/vittf.h).  Importing the package does not touch the GPU;
every compute entry point raises if the library or a gfx950 device is missing (no CPU fallback)."""
from . import _lib                                                  # noqa: F401
from ._lib import VittfError                                        # noqa: F401
from .weights import (ARCHS, synthetic_state_dict, load_state_dict_file, find_local_checkpoint,   # noqa: F401
                      fold_patch_embed, interpolate_pos_embed)
from .engine import HipViT                                          # noqa: F401
from .extract import (sizing, feature_volume, pooled_axis, k_slices, DeviceVolume, AXIS_DIMS)     # noqa: F401
from .similarity import sample_features3d, compute_similarities, assign_labels                    # noqa: F401
from .synthetic import synthetic_volume, ct_like_volume, shapes                                   # noqa: F401
from . import bilateral, samplers, scores, similarity, weights                         # noqa: F401

__all__ = ['HipViT', 'feature_volume', 'compute_similarities', 'sample_features3d', 'assign_labels',
           'synthetic_state_dict', 'sizing', 'VittfError']
