"""Build switches of the MI355X path as ONE frozen object, fixed at model construction.

`import_and_load(..., config=Config(...))` attaches the object to every sub-module of the network it builds
(`attach(model, cfg)`); the networks read `cfg(self).<switch>` and hand operator-level switches to the operator table as
keyword arguments (`CorrBlock(..., bwd_windows=...)`, `pwc_warp(..., deterministic=...)`, `dense_block(..., fused_masks=...)`,
`attention_softmax(..., gemm=...)`).  Nothing flips a module global after import: an A/B run builds a second model (the
weights are seeded, the packs are cached per tensor) with `dataclasses.replace(DEFAULT, switch=value)`.

Every switch selects between two executions of the same arithmetic (parity-tested against each other); the defaults are
the product path.  Environment variables only set the DEFAULT object, once, at import."""
import dataclasses
import os


def _env_bool(name, default):
    v = os.environ.get(name)
    return default if v is None else v not in ("0", "", "false", "False")


@dataclasses.dataclass(frozen=True)
class Config:
    # ---- RAFT / GMA (nets/raft.py, nets/gma.py) ----
    fused_lookup: bool = True            # lookup -> convc1 -> bias -> ReLU in one launch (ops.corr._CorrLookupConv)
    overlap_encoders: bool = False       # context encoder on a second stream beside the feature encoder (opt-in)
    conv_s2: bool = True                 # stride-2 layers on ops.conv_s2 (False: library convolution)
    conv_s2_bwd: bool = True             # their data gradient on pcfa_conv_s2_bwd (False: library gradient)
    fused_downsample: bool = True        # conv1 + downsample[0] of a stride-2 residual block in one launch
    defer_relu: bool = True              # ReLU backward of single-consumer layers inside neighbouring kernels
    pyramid_bwd_windows: bool = True     # pyramid backward over the lookup windows only (False: dense products)
    gma_gemm: str = "lib"                # GMA attention products: "lib" (rocBLAS through torch.matmul) | "hip" (pcfa_gemm_f32)
    conv1x1: str = "lib"                 # the encoders' output layer and the mask head's 1x1 layer: "lib" | "hip" (pcfa_gemm_f32:
                                         # a RAFT closure then holds no library kernel at all)
    # ---- PWC-Net (nets/pwcnet.py) ----
    dilated_as_subgrids: tuple = (2, 4, 8, 16)   # dilations run as d*d plain 3x3 convolutions on sub-grids (() = library)
    deconv_fewout: bool = True           # deconv / upfeat layers and the x4 up-sampling on own deterministic kernels
    defer_leaky: bool = True             # LeakyReLU backward of single-consumer layers in the consumer's epilogue
    dense_block_fused_masks: bool = True   # the same inside the dense decoder blocks
    warp_bwd_deterministic: bool = True  # fixed-point scatter in the warp backward (False: hardware fp32 atomics)
    pwc_fold_glue: bool = True           # RGB->BGR in conv1a's weights, `up_flow * s` inside the warp, decoder inputs written
                                         # straight into the dense-block buffer, one re-gridding copy between dilated layers
    # ---- attack loop (attack_PCFA.py) ----
    reuse_pair_graphs: bool = True       # pairs of one shape share static buffers + hipGraphs
    max_cached_shapes: int = 4           # graph sets kept per model (LRU); KITTI under /8 padding has three padded shapes

    @classmethod
    def from_env(cls):
        return cls(fused_lookup=_env_bool("PCFA_FUSED_LOOKUP", True),
                   overlap_encoders=_env_bool("PCFA_OVERLAP_ENCODERS", False),
                   defer_relu=_env_bool("PCFA_DEFER_RELU", True),
                   gma_gemm=os.environ.get("PCFA_GMA_GEMM", "lib"),
                   conv1x1=os.environ.get("PCFA_CONV1X1", "lib"),
                   max_cached_shapes=int(os.environ.get("PCFA_MAX_CACHED_SHAPES", "4")))

    def __post_init__(self):
        if self.gma_gemm not in ("lib", "hip"):
            raise ValueError("Config.gma_gemm must be 'lib' or 'hip', got %r" % (self.gma_gemm,))
        if self.conv1x1 not in ("lib", "hip"):
            raise ValueError("Config.conv1x1 must be 'lib' or 'hip', got %r" % (self.conv1x1,))
        if self.max_cached_shapes < 1:
            raise ValueError("Config.max_cached_shapes must be >= 1")


DEFAULT = Config.from_env()


def attach(model, config=None):
    """Give every sub-module of `model` the (frozen) config; returns the model."""
    config = DEFAULT if config is None else config
    if not isinstance(config, Config):
        raise TypeError("config must be a pcfa_amd.config.Config")
    for m in model.modules():
        object.__setattr__(m, "_pcfa_config", config)   # plain attribute: not a parameter / buffer / sub-module
    return model


def cfg(module):
    """The config a module was built with (DEFAULT for modules constructed on their own, e.g. in unit tests)."""
    return getattr(module, "_pcfa_config", DEFAULT)
