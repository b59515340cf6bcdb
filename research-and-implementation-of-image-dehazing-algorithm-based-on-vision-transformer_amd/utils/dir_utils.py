"""Drop-in for Uformer_ProbSparse/utils/dir_utils.py (natsort is not a dependency here: `natsorted` below orders
embedded integers numerically, which is what the reference needs it for - '10_2.png' after '9_12.png')."""
import os
import re
from glob import glob


def natsorted(seq):
    def key(s):
        return [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', str(s))]
    return sorted(seq, key=key)


def mkdir(path):
    if not os.path.exists(path):
        os.makedirs(path)


def mkdirs(paths):
    if isinstance(paths, (list, tuple)):
        for p in paths:
            mkdir(p)
    else:
        mkdir(paths)


def get_last_path(path, session):
    return natsorted(glob(os.path.join(path, '*%s' % session)))[-1]
