"""The operator table: every autograd-aware operator of the PCFA hot path, backed by libpcfa_hip.so.

This is the ONLY operator implementation inside the package.  Every op checks that its tensors live on a HIP device and
raises otherwise (no CPU fallback); launches go to torch's current stream through the C-ABI, so they order with the
surrounding library work and can be captured into a hipGraph.

Operator boundaries mirrored (reference file:line) -- one module per group:
  corr         CorrBlock                                      models/raft/corr.py:12-60 (== models/gma/corr.py:15-63)
  pwc          spatial_correlation_sample, pwc_cost_volume    .../spatial_correlation_sampler/spatial_correlation_sampler.py:9-91
               pwc_warp, deconv4s2_fewout, upsample_bilinear  models/PWCNet/PWCNet.py:166-206, :42-43, :73,321
  flownet      flownet_correlation, resample2d, channelnorm   models/FlowNet/{correlation,resample2d,channelnorm}_package/*.py
  conv         conv3x3, conv3x3_cat, dense_block, conv_s2(_ds), conv_fewin, conv3x3_fewout, sepconv5,
               instance_norm_relu, add_relu                   models/raft/update.py, extractor.py, PWCNet.py:29-38,234-323
  gru          gru_step, gru_gates(_packed), gru_update, bias_relu, fanout, flow_step, convex_upsample
                                                              models/raft/update.py:33-60, raft.py:72-83,122-137
  gma          attention_softmax, attn_times_value, gemm_f32  models/gma/gma.py:34-77,79-115
  attack_math  box_transform, extract_deltas(_joint), loss_delta_constraint, avg_epe, two_norm_*, pm1_pair
                                                              helper_functions/own_models.py:62-85, attack_PCFA.py:20-37, losses.py
"""
from .. import _hip  # noqa: F401
from ..lbfgs import LBFGS  # noqa: F401  (the attack loop's optimiser: torch.optim.LBFGS semantics, HIP vector math)
from .attack_math import *  # noqa: F401,F403
from .attack_math import _ExtractDeltas  # noqa: F401  (tests / tools reach a few private names through the table)
from .conv import *  # noqa: F401,F403
from .conv import _CONV_WS_RETIRED, _conv3x3_packed, _conv3x3_run, _conv_workspace, _sepconv5_packed  # noqa: F401
from .core import *  # noqa: F401,F403
from .core import _call, _dev, _ptr, _ptr_off, _stream  # noqa: F401
from .corr import *  # noqa: F401,F403
from .corr import _convc1_packed  # noqa: F401
from .flownet import *  # noqa: F401,F403
from .gma import *  # noqa: F401,F403
from .gru import *  # noqa: F401,F403
from .profiling import *  # noqa: F401,F403
from .pwc import *  # noqa: F401,F403
from .pwc import _PwcCostVolume  # noqa: F401
