"""The reference's own KATs (kat_scenarios.py) against the CPU oracle -- and, on
the GPU box, against the HIP path through the C ABI (`-m gpu`)."""
import pytest

import kat_scenarios as K
from oracle import oracle as orc


def _oracle(cfg):
    return orc.OracleEnv(cfg)


def test_square_kats_oracle():
    K.square_scenarios(_oracle)


def test_rect_kats_oracle():
    K.rect_scenarios(_oracle)


def test_pin_kats_oracle():
    K.pin_scenarios(_oracle, "pin")
    K.pin_scenarios(_oracle, "spatial")


def test_reward_chain_kats_oracle():
    K.reward_scenarios(orc)


def _gpu(cfg):
    from pcbenv.single_env import SingleEnvAdapter
    return SingleEnvAdapter(cfg)


@pytest.mark.gpu
def test_square_kats_hip():
    K.square_scenarios(_gpu)


@pytest.mark.gpu
def test_rect_kats_hip():
    K.rect_scenarios(_gpu)


@pytest.mark.gpu
def test_pin_kats_hip():
    K.pin_scenarios(_gpu, "pin")
    K.pin_scenarios(_gpu, "spatial")
