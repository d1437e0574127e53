import os

os.environ.setdefault("JV_DYNAMIC_ENV", "1")   # let tests flip JV_TILE / JV_OP_X6 / ... between calls (csrc/jv_common.h)
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def tts_sd():
    from jyutvoice_amd import synth
    return synth.tts_state_dict()


@pytest.fixture(scope="session")
def hift_sd():
    from jyutvoice_amd import synth
    return synth.hift_state_dict()


@pytest.fixture(scope="session")
def noise():
    from jyutvoice_amd import synth
    return synth.rand_noise()


@pytest.fixture(scope="session")
def prompt_sd():
    from jyutvoice_amd import synth
    return synth.prompt_state_dict()
