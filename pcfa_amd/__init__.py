"""pcfa_amd -- MI355X-native engine for the PCFA perturbation-optimisation hot path.

Host side (this package, PyTorch-ROCm for device memory / streams / convolutions /
torch.distributed) mirrors the reference's plugin API:

    pcfa_amd.helper_functions.ownutilities   import_and_load / preprocess_img / compute_flow / ...
    pcfa_amd.helper_functions.own_models     ScaledInputModel
    pcfa_amd.helper_functions.losses         loss_delta_constraint, avg_epe, ...
    pcfa_amd.helper_functions.targets        get_target
    pcfa_amd.attack_PCFA                     pcfa_attack, attack_l2, attack_l2_universal

The kernels it iterates are hand-written HIP behind the C-ABI of
include/pcfa_hip.h (pcfa_amd/lib/libpcfa_hip.so, sources in pcfa_amd/csrc).
"""
__version__ = "0.1.0"
