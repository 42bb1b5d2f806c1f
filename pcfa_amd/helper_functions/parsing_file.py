"""Command-line flags of the PCFA attack.

Flag names, defaults and choices are those of the reference's
helper_functions/parsing_file.py:3-98 (so existing command lines keep working); they are
declared here as a table.  Flags marked NEW configure what the reference takes from files
that do not exist in this environment: synthetic image pairs and seeded random weights.
"""
import argparse

_NETS = ['RAFT', 'GMA', 'PWCNet', 'SpyNet', 'FlowNet2']

# (group, flag, kwargs, condition) ; condition(stage, attack) -> bool
_ALWAYS = lambda s, a: True
_PCFA = lambda s, a: a == 'pcfa'
_PCFA_TRAIN = lambda s, a: a == 'pcfa' and s == 'training'
_PCFA_EVAL = lambda s, a: a == 'pcfa' and s == 'evaluation'
_FGSM = lambda s, a: a == 'fgsm'
_TRAIN = lambda s, a: s == 'training'

_FLAGS = [
    ("network", "--net", dict(default='SpyNet', choices=_NETS, help="flow network under attack"), _ALWAYS),
    ("network", "--weights", dict(default='pretrained', help="NEW: 'pretrained' (checkpoints in "
                                  "$PCFA_WEIGHTS_DIR) or 'random:<seed>'"), _ALWAYS),
    ("dataset", "--dataset", dict(default='Kitti15', choices=['Kitti15', 'Sintel', 'Synthetic'],
                                  help="image pairs to attack (NEW: 'Synthetic' = seeded pairs, no files)"), _ALWAYS),
    ("dataset", "--dataset_stage", dict(default='evaluation', choices=['training', 'evaluation'],
                                        help="dataset split"), _ALWAYS),
    ("dataset", "--small_run", dict(action='store_true', help="debug: first 32 samples only"), _ALWAYS),
    ("dataset", "--synthetic_size", dict(default='436x1024', help="NEW: HxW of synthetic pairs"), _ALWAYS),
    ("dataset", "--synthetic_pairs", dict(default=8, type=int, help="NEW: number of synthetic pairs"), _ALWAYS),
    ("sintel", "--dstype", dict(default='final', choices=['clean', 'final'], help="Sintel render pass"), _ALWAYS),
    ("saving", "--output_folder", dict(default='experiment_data', help="where artefacts and metrics go"), _ALWAYS),
    ("saving", "--small_save", dict(action='store_true', help="artefacts for the first 32 pairs only"), _ALWAYS),
    ("saving", "--save_frequency", dict(type=int, default=1, help="save artefacts every N-th pair"), _ALWAYS),
    ("saving", "--no_save", dict(action='store_true', help="write no artefacts at all"), _ALWAYS),
    ("saving", "--unregistered_artifacts", dict(action='store_true', default=False,
                                                help="kept for CLI compatibility (no artefact registry here)"),
     _ALWAYS),
    ("perturbation", "--joint_perturbation", dict(action='store_true', default=False,
                                                  help="one perturbation shared by both frames"), _ALWAYS),
    ("perturbation", "--steps", dict(default=20, type=int, help="optimisation steps per pair"), _ALWAYS),
    ("fgsm", "--epsilon", dict(default=0.00025, type=float, help="I-FGSM step size"), _FGSM),
    ("pcfa", "--universal_perturbation", dict(action='store_true', default=False,
                                              help="one perturbation for the whole dataset"), _PCFA),
    ("pcfa", "--boxconstraint", dict(default='change_of_variables', choices=['clipping', 'change_of_variables'],
                                     help="how images are kept inside [0,1]"), _PCFA),
    ("pcfa", "--batch_size", dict(default=4, type=int, help="[universal only] pairs per batch"), _PCFA),
    ("pcfa", "--pairs_in_flight", dict(default=1, type=int, help="NEW: independent pairs attacked side by side on every "
                                       "GPU (per-pair mode; results identical to 1, ~1.3x the pairs/s at 2)"), _PCFA_TRAIN),
    ("pcfa", "--delta_bound", dict(default=0.005, type=float, help="bound on the per-pixel averaged L2 norm of "
                                   "the perturbation"), _PCFA_TRAIN),
    ("pcfa", "--mu", dict(default=-1, type=float, help="penalty weight; -1 = 2500/delta_bound (x1.5 for non-zero "
                          "targets)"), _PCFA_TRAIN),
    ("pcfa", "--epochs", dict(default=25, type=int, help="[universal only] passes over the dataset"), _PCFA_TRAIN),
    ("pcfa", "--perturbation_sourcefolder", dict(help="folder or .npy with perturbations to evaluate"), _PCFA_EVAL),
    ("pcfa", "--origin_net", dict(help="network the perturbations were trained on"), _PCFA_EVAL),
    ("training", "--target", dict(default='zero', choices=['zero', 'neg_flow', 'custom'], help="target flow"), _TRAIN),
    ("training", "--custom_target_path", dict(default='', help="flow file for --target custom"), _TRAIN),
    ("training", "--loss", dict(default='aee', choices=['aee', 'mse', 'cosim'], help="similarity term"), _TRAIN),
]


def create_parser(stage=None, attack_type=None):
    stage = stage.lower()
    attack_type = attack_type.lower()
    if stage not in ['training', 'evaluation']:
        raise ValueError('To create a parser the stage has to be specified. Please choose one of "training" or '
                         '"evaluation"')
    if attack_type not in ["fgsm", "pcfa"]:
        raise ValueError('To create a parser the attack type has to be specified. Please choose one of "fgsm" or '
                         '"pcfa"')
    parser = argparse.ArgumentParser(usage='%(prog)s [options (see below)]')
    groups = {}
    for group, flag, kwargs, cond in _FLAGS:
        if not cond(stage, attack_type):
            continue
        if group not in groups:
            groups[group] = parser.add_argument_group(title="%s arguments" % group)
        groups[group].add_argument(flag, **kwargs)
    return parser
