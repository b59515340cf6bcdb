"""Drop-in for Uformer_ProbSparse/utils/loader.py: the four dataset factories, each guarding that its directory exists."""
import os

import dataset as _ds


def _checked(cls, rgb_dir, *args):
    if not os.path.exists(rgb_dir):
        raise AssertionError(f"data directory not found: {rgb_dir}")
    return cls(rgb_dir, *args, None)


def get_training_data(rgb_dir, img_options):
    """(clean, noisy, clean_filename, noisy_filename) items with random crop + augmentation."""
    return _checked(_ds.DataLoaderTrain, rgb_dir, img_options)


def get_validation_data(rgb_dir):
    return _checked(_ds.DataLoaderVal, rgb_dir)


def get_test_data(rgb_dir):
    return _checked(_ds.DataLoaderTest, rgb_dir)


def get_test_data_SR(rgb_dir):
    return _checked(_ds.DataLoaderTestSR, rgb_dir)
