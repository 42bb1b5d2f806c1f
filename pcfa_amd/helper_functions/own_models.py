"""ScaledInputModel (reference helper_functions/own_models.py:9-88).

Wraps a flow network so that it accepts the attack's optimisation variables: adds the
perturbation, applies the box constraint (tanh change of variables or clipping), rescales
[0,1] -> [0,255] where the network wants it, then delegates to compute_flow.  The
elementwise chain runs as one fused HIP kernel per image (box_transform in
pcfa_amd/csrc/attack_math.hip) instead of ~8 separate elementwise launches.
"""
import logging

import torch.nn as nn

from .. import ops
from . import ownutilities


class ScaledInputModel(nn.Module):
    def __init__(self, net, make_unit_input=False, variable_change=False, **kwargs):
        super().__init__()
        self.make_unit_input = make_unit_input
        self.var_change = variable_change
        self.model_name = net
        logging.info("Creating a Model with scaled input and the following parameters:")
        logging.info("\tmake_unit_input=%s" % (str(make_unit_input)))
        self.eps_box = 0.
        if 'eps_box' in kwargs:
            self.eps_box = kwargs.pop("eps_box")
            logging.info("\teps_box=%s" % (str(self.eps_box)))
        elif variable_change:
            logging.warning("The ScaledInputModel did receive 'variable_change'=True, but no epsilon value for the "
                            "CW attack was given. This might lead to numerical instabilities. Consider a small "
                            "float value to the ScaledInputModel. Setting 'eps_box'=0.0.")
        self.model_loaded = ownutilities.import_and_load(net, **kwargs)

    def forward(self, image1, image2, delta1=None, delta2=None, test_mode=True, *args, **kwargs):
        """own_models.py:40-88: `delta1` alone perturbs both images; `delta1` and `delta2` perturb one each."""
        d1 = delta1
        d2 = delta2 if delta2 is not None else delta1
        scale = 255. if self.make_unit_input else 1.
        box = ops.get().box_transform
        image1 = box(image1, d1, self.var_change, self.eps_box, scale)
        image2 = box(image2, d2, self.var_change, self.eps_box, scale)
        return ownutilities.compute_flow(self.model_loaded, self.model_name, image1, image2, test_mode=test_mode,
                                         *args, **kwargs)
