"""Name -> class registries with the reference's plugin contract.

Contract restated from lightx2v/utils/registry_factory.py:1-56: a registry is called as a decorator
either bare (`@REG`, key = class name) or with a key (`@REG("name")`); registering a key twice raises
`Exception`; looking up an unknown key raises `KeyError`; `in`, `keys()`, `values()`, `items()` work.
The instances carry the reference's names so code written against it reads the same here.

`adopt_into(reference_module)` copies this package's HIP plugins into the reference's own registry
objects, which is how a LightX2V checkout selects them from its JSON configs.
"""

from __future__ import annotations

from typing import Callable, Dict, Iterator, Optional


class Register:
    def __init__(self, label: str = ""):
        self.label = label
        self._entries: Dict[str, Callable] = {}

    # decorator forms -------------------------------------------------------------------------
    def __call__(self, target_or_name):
        if callable(target_or_name):
            return self.register(target_or_name)

        def _decorate(target):
            return self.register(target, key=target_or_name)

        return _decorate

    def register(self, target: Callable, key: Optional[str] = None):
        if not callable(target):
            raise Exception(f"Error: {target} must be callable!")
        name = key if key is not None else target.__name__
        if name in self._entries:
            raise Exception(f"{name} already exists.")
        self._entries[name] = target
        return target

    # mapping protocol ------------------------------------------------------------------------
    def __getitem__(self, key: str) -> Callable:
        return self._entries[key]

    def __setitem__(self, key: str, value: Callable):
        self._entries[key] = value

    def __contains__(self, key) -> bool:
        return key in self._entries

    def __iter__(self) -> Iterator[str]:
        return iter(self._entries)

    def __len__(self) -> int:
        return len(self._entries)

    def __str__(self) -> str:
        return str(self._entries)

    def keys(self):
        return self._entries.keys()

    def values(self):
        return self._entries.values()

    def items(self):
        return self._entries.items()


MM_WEIGHT_REGISTER = Register("mm")
ATTN_WEIGHT_REGISTER = Register("attn")
RMS_WEIGHT_REGISTER = Register("rms")
LN_WEIGHT_REGISTER = Register("ln")
TENSOR_REGISTER = Register("tensor")
CONV3D_WEIGHT_REGISTER = Register("conv3d")
EMBEDDING_WEIGHT_REGISTER = Register("embedding")

_ALL = {
    "MM_WEIGHT_REGISTER": MM_WEIGHT_REGISTER,
    "ATTN_WEIGHT_REGISTER": ATTN_WEIGHT_REGISTER,
    "RMS_WEIGHT_REGISTER": RMS_WEIGHT_REGISTER,
    "LN_WEIGHT_REGISTER": LN_WEIGHT_REGISTER,
    "TENSOR_REGISTER": TENSOR_REGISTER,
    "CONV3D_WEIGHT_REGISTER": CONV3D_WEIGHT_REGISTER,
    "EMBEDDING_WEIGHT_REGISTER": EMBEDDING_WEIGHT_REGISTER,
}


def adopt_into(reference_registry_module, only_prefix: str = "hip") -> Dict[str, list]:
    """Register every plugin of this package whose key starts with `only_prefix` into the
    same-named registries of `reference_registry_module` (the reference's
    `lightx2v.utils.registry_factory`).  Returns {registry name: [keys added]}."""
    added: Dict[str, list] = {}
    for reg_name, mine in _ALL.items():
        theirs = getattr(reference_registry_module, reg_name, None)
        if theirs is None:
            continue
        for key, cls in mine.items():
            if key.startswith(only_prefix) and key not in theirs:
                theirs.register(cls, key=key)
                added.setdefault(reg_name, []).append(key)
    return added
