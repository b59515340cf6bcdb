"""Drop-in for Uformer_ProbSparse/options.py: same flags and defaults (options.py:13-73) and the module
level ablation switch `is_relative_position_bias` (options.py:5) that the attention reads at call time
(ATT:227).  Paths that were machine-specific in the reference default to relative locations.

The flag set is data: (name, type or 'switch', default).  tests/test_host.py checks names and defaults against the
reference's own parser (tests/golden/options.npz)."""

is_relative_position_bias = True          # ablation switch: relative position bias inside the window attention

SWITCH = 'switch'                         # argparse store_true flag, default False

_FLAGS = (
    # loss mix ('is_ab' keeps the reference quirk: type=bool, so any non-empty string parses as True)
    ('is_ab', bool, False), ('w_loss_vgg7', float, 1), ('w_loss_CharbonnierLoss', float, 1),
    # run
    ('batch_size', int, 32), ('nepoch', int, 250), ('train_workers', int, 12), ('eval_workers', int, 8),
    ('dataset', str, 'Dense-HAZE'), ('pretrain_weights', str, './log/UformerResave_all_My_Infor_CR/models/model_best.pth'),
    ('optimizer', str, 'adamw'), ('lr_initial', float, 0.0002), ('weight_decay', float, 0.02), ('gpu', str, '0,1'),
    ('arch', str, 'Uformer'), ('mode', str, 'denoising'),
    # output
    ('save_dir', str, '/home/ma-user/work/deNoTr/log'), ('save_images', SWITCH, False), ('env', str, '_'), ('checkpoint', int, 50),
    # network
    ('norm_layer', str, 'nn.LayerNorm'), ('embed_dim', int, 32), ('win_size', int, 8), ('token_projection', str, 'linear'),
    ('token_mlp', str, 'leff'), ('att_se', SWITCH, False),
    # ViT bottleneck flags: unused by the Uformer path, kept so that reference command lines still parse
    ('vit_dim', int, 256), ('vit_depth', int, 12), ('vit_nheads', int, 8), ('vit_mlp_dim', int, 512), ('vit_patch_size', int, 16),
    ('global_skip', SWITCH, False), ('local_skip', SWITCH, False), ('vit_share', SWITCH, False),
    # training data / schedule
    ('train_ps', int, 128), ('resume', SWITCH, False), ('train_dir', str, '../datasets/SIDD/train'),
    ('val_dir', str, '../datasets/SIDD/val'), ('warmup', SWITCH, False), ('warmup_epochs', int, 3),
)


class Options():
    def __init__(self):
        pass

    def init(self, parser):
        for name, kind, default in _FLAGS:
            if kind == SWITCH:
                parser.add_argument('--' + name, action='store_true', default=False)
            else:
                parser.add_argument('--' + name, type=kind, default=default)
        return parser
