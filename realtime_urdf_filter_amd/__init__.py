"""MI355X-native realtime URDF depth self-filter (hot path of blodow/realtime_urdf_filter)."""
from ._capi import (Context, Params, RtufError, default_params, load_library,  # noqa: F401
                    projection_from_intrinsics, OP_NONE, OP_SCALE, OP_TRANSLATE,
                    FLAG_TWO_KERNEL, FLAG_STRICT_GRID, ABI_VERSION, expand_mask_bits,
                    STATUS_PENDING_MASK, STATUS_BIN_OVERFLOW, STATUS_CLIP_OVERFLOW, STATUS_LIST_OVERFLOW,
                    STATUS_GRID_SHORT, STATUS_UNCOVERED)
