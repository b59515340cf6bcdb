"""Drop-in for Uformer_ProbSparse/utils: the symbols My_train.py uses (get_arch, load_checkpoint*,
load_start_epoch, load_optim, MixUp_AUG, mkdir, PSNR helpers)."""
from .dir_utils import *      # noqa: F401,F403
from .dataset_utils import *  # noqa: F401,F403
from .image_utils import *    # noqa: F401,F403
from .model_utils import *    # noqa: F401,F403
