"""Ordered container of named blocks with the reference's vectorisation rules
(optas/sx_container.py): ``vec()`` order = insertion order, each block column-major (:83-89);
``dict2vec`` zero-fills missing labels (:113-123); ``vec2dict`` reshapes column-major (:98-111);
duplicate labels raise ``KeyError`` (:50-51).  Values are ``Expr`` nodes instead of ``casadi.SX``.
"""
from __future__ import annotations

import collections
from typing import Dict, List

import numpy as np

from .expr import Expr


class SXContainer(collections.OrderedDict):
    """Contract mirrored from the reference class (what callers and the reference's tests rely on): insertion order is vec() order; a label can be
    set once (KeyError afterwards, sx_container.py:50-51); every block starts out continuous and can be flagged discrete by label; ``a + b`` is a
    new container holding a's blocks, then b's, with both flag tables."""

    def __init__(self):
        super().__init__()
        self.is_discrete: Dict[str, bool] = {}

    def _adopt(self, other: "SXContainer") -> None:
        for label in other:
            self[label] = other[label]
            self.is_discrete[label] = other.is_discrete.get(label, False)

    def __add__(self, other):
        if not isinstance(other, SXContainer):
            raise AssertionError(f"an SXContainer can only be joined with another one, got {type(other).__name__}")
        joined = SXContainer()
        joined._adopt(self)
        joined._adopt(other)
        return joined

    def __setitem__(self, label: str, value: Expr) -> None:
        if not isinstance(value, Expr):
            raise AssertionError(f"blocks are optas_amd expressions; '{label}' was given a {type(value).__name__}")
        if label in self:
            raise KeyError(f"'{label}' already exists")
        collections.OrderedDict.__setitem__(self, label, value)
        self.is_discrete.setdefault(label, False)

    def variable_is_discrete(self, label: str) -> None:
        if label not in self:
            raise AssertionError(f"no block is called '{label}'")
        self.is_discrete[label] = True

    def has_discrete_variables(self) -> bool:
        return True in self.is_discrete.values()

    def discrete(self) -> List[bool]:
        """One flag per scalar entry, in vec() order."""
        return [flag for label, block in self.items() for flag in [self.is_discrete[label]] * (block.shape[0] * block.shape[1])]

    def numel(self) -> int:
        return sum(v.shape[0] * v.shape[1] for v in self.values())

    def offsets(self) -> Dict[str, int]:
        off, out = 0, {}
        for label, value in self.items():
            out[label] = off
            off += value.shape[0] * value.shape[1]
        return out

    def vec2dict(self, vec) -> dict:
        vec = np.asarray(vec, dtype=np.float64).reshape(-1)
        out, off = {}, 0
        for label, value in self.items():
            m, n = value.shape
            out[label] = vec[off : off + m * n].reshape(n, m).T.copy()
            off += m * n
        return out

    def dict2vec(self, d: Dict[str, np.ndarray]) -> np.ndarray:
        parts = []
        for label, value in self.items():
            m, n = value.shape
            v = d.get(label)
            if v is None:
                parts.append(np.zeros(m * n))
            else:
                a = np.asarray(v, dtype=np.float64)
                if a.ndim <= 1:
                    a = a.reshape(-1, 1) if a.size == m * n and n == 1 else a.reshape(m, n)
                assert a.shape == (m, n), f"'{label}' expects shape {(m, n)}, got {a.shape}"
                parts.append(a.T.reshape(-1))
        return np.concatenate(parts) if parts else np.zeros(0)

    def dict2vec_batch(self, d: Dict[str, np.ndarray], B: int) -> np.ndarray:
        """Batched dict2vec: every value carries a leading batch axis (B, m, n) (or (B, m) / (B,) for columns / scalars); missing
        labels are zero-filled like dict2vec.  Returns (B, numel) rows in vec() order."""
        out = np.zeros((B, self.numel()))
        off = 0
        for label, value in self.items():
            m, n = value.shape
            v = d.get(label)
            if v is not None:
                a = np.asarray(v, dtype=np.float64)
                assert a.shape[0] == B, f"'{label}': leading axis must be the batch size {B}"
                a = a.reshape(B, m, n) if a.size == B * m * n else None
                assert a is not None, f"'{label}' expects per-instance shape {(m, n)}"
                out[:, off : off + m * n] = a.transpose(0, 2, 1).reshape(B, m * n)
            off += m * n
        return out

    def vec2dict_batch(self, vecs) -> Dict[str, np.ndarray]:
        """(B, numel) -> {label: (B, m, n)}."""
        vecs = np.asarray(vecs, dtype=np.float64)
        B = vecs.shape[0]
        out, off = {}, 0
        for label, value in self.items():
            m, n = value.shape
            out[label] = vecs[:, off : off + m * n].reshape(B, n, m).transpose(0, 2, 1).copy()
            off += m * n
        return out

    def zero(self) -> dict:
        return {label: np.zeros(value.shape) for label, value in self.items()}
