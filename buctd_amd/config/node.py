"""A small yacs-compatible configuration node (yacs is not installed in the target image).

Supports what the BUCTD path uses (reference lib/config/default.py:180-207, tools/train.py:77-94):
attribute and item access, nested nodes, merge_from_file (YAML), merge_from_list (KEY VALUE ... with
yacs-style literal parsing), new_allowed sub-trees (MODEL.EXTRA), freeze / defrost, clone.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"
    NEW_ALLOWED = "__new_allowed__"

    def __init__(self, init_dict=None, new_allowed=False):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        self.__dict__[CfgNode.NEW_ALLOWED] = new_allowed
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v, new_allowed=new_allowed) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_frozen(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode(new_allowed=self.__dict__[CfgNode.NEW_ALLOWED])
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        return out

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @staticmethod
    def _coerce(new, old, key):
        if old is None or new is None or type(new) is type(old):
            return new
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        if isinstance(old, bool) and isinstance(new, str) and new in ("True", "False"):
            return new == "True"
        raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(new), key))

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k in self:
                if isinstance(self[k], CfgNode) and isinstance(v, dict):
                    self[k]._merge(v, path + [k])
                else:
                    dict.__setitem__(self, k, self._coerce(self._decode(v) if not isinstance(v, dict) else v,
                                                            self[k], full))
            elif self.__dict__[CfgNode.NEW_ALLOWED]:
                dict.__setitem__(self, k, CfgNode(v, new_allowed=True) if isinstance(v, dict) else v)
            else:
                raise KeyError("Non-existent config key: {}".format(full))

    def merge_from_file(self, path):
        with open(path, "r") as f:
            self._merge(yaml.safe_load(f) or {}, [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_list(self, opts):
        if len(opts) % 2 != 0:
            raise ValueError("Override list has odd length: {}; it must be a list of pairs".format(opts))
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent key: {}".format(key))
                node = node[p]
            leaf = parts[-1]
            val = self._decode(val)
            if leaf in node:
                dict.__setitem__(node, leaf, self._coerce(val, node[leaf], key))
            elif node.__dict__[CfgNode.NEW_ALLOWED]:
                dict.__setitem__(node, leaf, val)
            else:
                raise KeyError("Non-existent key: {}".format(key))
