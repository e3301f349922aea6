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
    def __init__(self):
        super().__init__()
        self.is_discrete: Dict[str, bool] = {}

    def __add__(self, other):
        assert isinstance(other, SXContainer), f"cannot add SXContainer with a variable of type {type(other)}"
        out = SXContainer()
        for label, value in self.items():
            out[label] = value
        for label, value in other.items():
            out[label] = value
        out.is_discrete = {**self.is_discrete, **other.is_discrete}
        return out

    def __setitem__(self, label: str, value: Expr) -> None:
        assert isinstance(value, Expr), f"value must be an optas_amd expression, not {type(value)}"
        if label in self:
            raise KeyError(f"'{label}' already exists")
        super().__setitem__(label, value)
        self.is_discrete[label] = False

    def variable_is_discrete(self, label: str) -> None:
        assert label in self, f"'{label}' was not found"
        self.is_discrete[label] = True

    def has_discrete_variables(self) -> bool:
        return any(self.is_discrete.values())

    def discrete(self) -> List[bool]:
        out: List[bool] = []
        for label, value in self.items():
            m, n = value.shape
            out += [self.is_discrete[label]] * (m * n)
        return out

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
