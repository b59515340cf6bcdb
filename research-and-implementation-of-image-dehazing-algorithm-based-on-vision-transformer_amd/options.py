"""Drop-in for Uformer_ProbSparse/options.py: same flags and defaults (options.py:13-73) and the module
level ablation switch `is_relative_position_bias` (options.py:5) that the attention reads at call time
(ATT:227).  Paths that were machine-specific in the reference default to relative locations."""
import os  # noqa: F401

######## Ablation Study ########
is_relative_position_bias = True


class Options():
    def __init__(self):
        pass

    def init(self, parser):
        add = parser.add_argument
        # loss mix
        add('--is_ab', type=bool, default=False)            # reference quirk kept: any non-empty string is True
        add('--w_loss_vgg7', type=float, default=1)
        add('--w_loss_CharbonnierLoss', type=float, default=1)
        # global settings
        add('--batch_size', type=int, default=32, help='batch size')
        add('--nepoch', type=int, default=250, help='training epochs')
        add('--train_workers', type=int, default=12, help='train_dataloader workers')
        add('--eval_workers', type=int, default=8, help='eval_dataloader workers')
        add('--dataset', type=str, default='Dense-HAZE')
        add('--pretrain_weights', type=str, default='./log/UformerResave_all_My_Infor_CR/models/model_best.pth',
            help='path of pretrained_weights')
        add('--optimizer', type=str, default='adamw', help='optimizer for training')
        add('--lr_initial', type=float, default=0.0002, help='initial learning rate')
        add('--weight_decay', type=float, default=0.02, help='weight decay')
        add('--gpu', type=str, default='0,1', help='GPUs')
        add('--arch', type=str, default='Uformer', help='archtechture')
        add('--mode', type=str, default='denoising', help='image restoration mode')
        # saving
        add('--save_dir', type=str, default='/home/ma-user/work/deNoTr/log', help='save dir')
        add('--save_images', action='store_true', default=False)
        add('--env', type=str, default='_', help='env')
        add('--checkpoint', type=int, default=50, help='checkpoint')
        # Uformer
        add('--norm_layer', type=str, default='nn.LayerNorm', help='normalize layer in transformer')
        add('--embed_dim', type=int, default=32, help='dim of emdeding features')
        add('--win_size', type=int, default=8, help='window size of self-attention')
        add('--token_projection', type=str, default='linear', help='linear/convoptimizer token projection')
        add('--token_mlp', type=str, default='leff', help='ffn/leff token mlp')
        add('--att_se', action='store_true', default=False, help='se after sa')
        # vit (unused by the Uformer path; kept so reference command lines still parse)
        add('--vit_dim', type=int, default=256, help='vit hidden_dim')
        add('--vit_depth', type=int, default=12, help='vit depth')
        add('--vit_nheads', type=int, default=8, help='vit hidden_dim')
        add('--vit_mlp_dim', type=int, default=512, help='vit mlp_dim')
        add('--vit_patch_size', type=int, default=16, help='vit patch_size')
        add('--global_skip', action='store_true', default=False, help='global skip connection')
        add('--local_skip', action='store_true', default=False, help='local skip connection')
        add('--vit_share', action='store_true', default=False, help='share vit module')
        # training
        add('--train_ps', type=int, default=128, help='patch size of training sample')
        add('--resume', action='store_true', default=False)
        add('--train_dir', type=str, default='../datasets/SIDD/train', help='dir of train data')
        add('--val_dir', type=str, default='../datasets/SIDD/val', help='dir of train data')
        add('--warmup', action='store_true', default=False, help='warmup')
        add('--warmup_epochs', type=int, default=3, help='epochs for warmup')
        return parser
