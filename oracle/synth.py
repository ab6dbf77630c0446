"""Synthetic weights / images / predictions / assigner inputs.  The generators live in the
product package (yolov6_amd/utils/synth.py: bench.py fills its model with them); the oracle and
the golden generator use the same functions so every side sees bit-identical data."""
from yolov6_amd.utils.synth import *  # noqa: F401,F403
from yolov6_amd.utils.synth import (synth_images, synth_predictions, synth_state_dict, synth_tal_inputs,  # noqa: F401
                                    synth_tensor)
