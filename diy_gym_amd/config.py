"""YAML-backed configuration tree.

Host-side counterpart of the reference's ``Configuration`` (reference:
diy_gym/config.py:13-61).  Same surface -- ``from_file``, ``get(key, default)``,
``set``, ``in``, ``find_all``, ``find`` -- and the same error behaviour
(``KeyError`` for a missing key without a default).  One deliberate difference:
a YAML parse error is re-raised instead of being printed and then crashing with
a ``NameError`` (reference config.py:17-22).
"""
import os

import yaml

_MISSING = object()


class Configuration:
    def __init__(self, name, node, source_dir=''):
        self.name = name
        self.node = node if node is not None else {}
        self.source_dir = source_dir  # directory of the YAML file (relative URDF paths resolve against it)

    @classmethod
    def from_file(cls, path):
        with open(path, 'r') as fh:
            tree = yaml.load(fh, Loader=yaml.FullLoader)
        if tree is None:
            tree = {}
        stem = os.path.splitext(os.path.basename(path))[0]
        return cls(tree.get('name', stem), tree, os.path.dirname(os.path.abspath(path)))

    @classmethod
    def from_dict(cls, name, tree):
        return cls(tree.get('name', name), tree)

    def get(self, key, default=_MISSING):
        if key in self.node:
            value = self.node[key]
            return Configuration(key, value) if isinstance(value, dict) else value
        if default is _MISSING:
            raise KeyError("Couldn't find config and no default provided for config with key: " + str(key))
        return default

    def set(self, key, value):
        self.node[key] = value

    def __contains__(self, key):
        return key in self.node

    def find_all(self, key):
        """Direct children that are mappings containing ``key`` (YAML order)."""
        for child_name, child in self.node.items():
            if isinstance(child, dict) and key in child:
                yield Configuration(child_name, child)

    def find(self, key):
        return next(iter(self.find_all(key)))

    def __repr__(self):
        return 'Configuration(%r, keys=%r)' % (self.name, list(self.node.keys()))
