"""Minimal stand-in for the reference's hydra configs (hydra / omegaconf are not installed here):
PyYAML files with the reference's keys (``configs/train.yaml:5-24``, ``configs/predict.yaml:5-14``),
``defaults: [main, _self_]`` composition and ``key=value`` command-line overrides."""
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs')


def _coerce(v):
    try:
        return yaml.safe_load(v)
    except yaml.YAMLError:
        return v


def load_config(name, overrides=(), config_dir=CONFIG_DIR):
    with open(os.path.join(config_dir, f'{name}.yaml')) as f:
        cfg = yaml.safe_load(f) or {}
    merged = {}
    for d in cfg.pop('defaults', []):
        if d != '_self_' and os.path.exists(os.path.join(config_dir, f'{d}.yaml')):
            merged.update(load_config(d, config_dir=config_dir))
    merged.update(cfg)
    for ov in overrides:
        k, _, v = ov.partition('=')
        node = merged
        parts = k.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _coerce(v)
    return merged
