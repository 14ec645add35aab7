"""Seeded synthetic checkpoints / inputs for the parity tests.

The generator itself is plain data synthesis (no model arithmetic) and lives in the package so
that bench.py does not depend on oracle/; re-exported here for the tests and gen_golden.py."""
from wsi_segmentation_pipeline_amd.synthetic import (make_he_patches, make_head_state_dict,  # noqa: F401
                                                     make_resnet18_state_dict, make_u8_patches, make_unet_state_dict,
                                                     make_wide_resnet18_state_dict, resnet18_key_shapes)
