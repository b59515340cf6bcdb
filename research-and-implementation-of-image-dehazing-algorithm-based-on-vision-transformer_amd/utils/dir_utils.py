import os


def mkdir(path):
    if not os.path.exists(path):
        os.makedirs(path)


def mkdirs(paths):
    if isinstance(paths, (list, tuple)):
        for p in paths:
            mkdir(p)
    else:
        mkdir(paths)
