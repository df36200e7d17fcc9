"""Module tree that owns op objects and fans out load / state_dict / device moves.

Same surface as the reference's container (lightx2v/common/modules/weight_module.py:1-196):
`add_module`, `register_parameter`, `load(weight_dict)`, `state_dict`, `load_state_dict`,
`named_parameters`, `to_cuda/to_cpu[_async]`, and a list flavour indexable by position.
Children only need the methods they implement; missing ones are skipped.
"""

from __future__ import annotations

from typing import Any, Dict, Iterable, Iterator, Optional, Tuple


class WeightModule:
    def __init__(self):
        self._modules: Dict[str, Any] = {}
        self._parameters: Dict[str, Any] = {}
        self._fused: Dict[str, Any] = {}

    # ---- tree construction --------------------------------------------------------------------
    def add_module(self, name: str, module: Any):
        self._modules[name] = module
        setattr(self, name, module)

    def register_parameter(self, name: str, param: Any):
        self._parameters[name] = param
        setattr(self, name, param)

    def is_empty(self) -> bool:
        return not self._modules and not self._parameters

    def _children(self) -> Iterable[Any]:
        yield from (m for m in self._modules.values() if m is not None)
        yield from (p for p in self._parameters.values() if p is not None)

    def _fanout(self, method: str, *args, **kwargs):
        for child in self._children():
            fn = getattr(child, method, None)
            if fn is not None:
                fn(*args, **kwargs)

    # ---- fused views ----------------------------------------------------------------------------
    def fused(self, key: str, members):
        """One Linear plugin over the row-wise concatenation of `members`' weights (q|k|v, k|v): one
        GEMM instead of several.  Built on first use; the members are then re-pointed at slices of
        the fused storage, so there is one copy.  The cached plugin is dropped whenever the
        members' weights can change (`load`, `load_state_dict`, `clear`, device moves) and is
        rebuilt if a member no longer lives in its storage."""
        cached = self._fused.get(key)
        if cached is not None:
            base = cached.weight_nk.untyped_storage().data_ptr()
            if all(m.weight_nk is not None and m.weight_nk.untyped_storage().data_ptr() == base for m in members):
                return cached
        built = type(members[0]).fused(members)
        self._fused[key] = built
        return built

    # ---- weights ------------------------------------------------------------------------------
    def load(self, weight_dict: Dict[str, Any]):
        self._fused.clear()
        self._fanout("load", weight_dict)

    def load_from_disk(self):
        self._fanout("load_from_disk")

    def clear(self):
        self._fused.clear()
        self._fanout("clear")

    def calculate_size(self) -> int:
        total = 0
        for child in self._children():
            if isinstance(child, WeightModule):
                total += child.calculate_size()
            elif hasattr(child, "_calculate_size"):
                total += child._calculate_size()
        return total

    def state_dict(self, destination: Optional[dict] = None) -> dict:
        if destination is None:
            destination = {}
        for p in self._parameters.values():
            if p is not None:
                p.state_dict(destination)
        for m in self._modules.values():
            if m is not None:
                m.state_dict(destination)
        return destination

    def load_state_dict(self, destination: Optional[dict], block_index: int, adapter_block_index: Optional[int] = None) -> dict:
        if destination is None:
            destination = {}
        self._fused.clear()
        for child in list(self._parameters.values()) + list(self._modules.values()):
            if child is not None:
                child.load_state_dict(destination, block_index, adapter_block_index)
        return destination

    def named_parameters(self, prefix: str = "") -> Iterator[Tuple[str, Any]]:
        for name, p in self._parameters.items():
            if p is not None:
                yield prefix + name, p
        for name, m in self._modules.items():
            if m is None:
                continue
            if isinstance(m, WeightModule):
                yield from m.named_parameters(prefix + name + ".")
            else:
                yield prefix + name, m

    # ---- device moves -------------------------------------------------------------------------
    def _move(self, method: str, non_blocking: bool):
        self._fused.clear()  # the members are moved one by one and stop aliasing the fused storage
        for child in self._children():
            if isinstance(child, WeightModule):
                child._move(method, non_blocking)
            else:
                fn = getattr(child, method, None)
                if fn is not None:
                    fn(non_blocking=non_blocking)

    def to_cuda(self, non_blocking: bool = False):
        self._move("to_cuda", non_blocking)

    def to_cpu(self, non_blocking: bool = False):
        self._move("to_cpu", non_blocking)

    def to_cuda_async(self):
        self._move("to_cuda", True)

    def to_cpu_async(self):
        self._move("to_cpu", True)


class WeightModuleList(WeightModule):
    def __init__(self, modules: Optional[Iterable[Any]] = None):
        super().__init__()
        self._list = []
        for m in modules or ():
            self.append(m)

    def append(self, module: Any):
        self.add_module(str(len(self._list)), module)
        self._list.append(module)

    def __getitem__(self, idx):
        return self._list[idx]

    def __setitem__(self, idx: int, module: Any):
        self._list[idx] = module
        self.add_module(str(idx), module)

    def __len__(self) -> int:
        return len(self._list)

    def __iter__(self):
        return iter(self._list)
