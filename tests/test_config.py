"""Configuration semantics, restating the reference's diy_gym/tests/test_config.py:11-23
on the reference's own fixture (tests/golden/basic_env.yaml)."""
import os

import pytest

from diy_gym_amd.config import Configuration

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'basic_env.yaml')


@pytest.fixture
def config():
    return Configuration.from_file(FIXTURE)


def test_get_config(config):
    assert config.get('im_a_config') is True


def test_default_config(config):
    assert config.get('im_not_a_config', False) is False


def test_missing_key_raises_keyerror(config):
    with pytest.raises(KeyError):
        config.get('im_not_a_config')


def test_set_config(config):
    config.set('im_a_config_now_too', 5.0)
    assert config.get('im_a_config_now_too') == 5.0


def test_find_all(config):
    assert len(list(config.find_all('model'))) == 4
    assert [c.name for c in config.find_all('addon')] == ['camera']


def test_name_defaults_to_file_stem(config):
    assert config.name == 'basic_env'
    assert 'render' in config and config.get('render') is False  # YAML 1.1 `no`


def test_nested_get_returns_configuration(config):
    sub = config.get('blue_marble')
    assert isinstance(sub, Configuration) and sub.get('model') == 'sphere2.urdf'
    assert sub.find('addon').name == 'force'
