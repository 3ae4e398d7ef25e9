"""Model -> tableau (host side).  Restates the reference's `tableauModel`
(/root/reference/src/tableau.ts:47-137) and the Tableau layout (:9-21): it feeds the HIP core,
it is not part of the accelerated path.  Pinned by tests/test_host_model.py against the initial
tableaux the reference itself built for its 46 test models (tests/golden/simplex_cases.json.gz).

A model is a dict shaped like the reference's `Model` (src/types.ts:48-148):
  {"direction": "maximize"|"minimize", "objective": key,
   "constraints": {key: {"equal"|"min"|"max": number}} or iterable of (key, constraint),
   "variables":   {key: {constraint_key: coef}}        or iterable of (key, coefficients),
   "integers": bool | iterable of variable keys, "binaries": bool | iterable of variable keys}
"""
import math
from dataclasses import dataclass

import numpy as np


@dataclass
class Tableau:
    """src/tableau.ts:9-15.  matrix is row-major width*height float64; row 0 = objective row,
    column 0 = RHS column; the two permutations have width+height int32 entries."""
    matrix: np.ndarray
    width: int
    height: int
    position_of_variable: np.ndarray
    variable_at_position: np.ndarray
    col0: np.ndarray = None  # RHS column alone, when only that was copied back from the GPU (matrix may be None)
    cells: tuple = None  # (row, col, val) of the cells tableauModel wrote, sorted by (row, col), when the
    #                      tableau was built sparse (matrix is None until dense() or the GPU fills col0)

    def dense(self):
        """The row-major matrix of a tableau built with sparse=True."""
        if self.matrix is None:
            row, col, val = self.cells
            self.matrix = np.zeros(self.width * self.height, np.float64)
            self.matrix[row.astype(np.int64) * self.width + col] = val
        return self.matrix

    def rhs(self, row):
        """index(tableau, row, 0)"""
        return float(self.col0[row]) if self.col0 is not None else float(self.matrix[row * self.width])


@dataclass
class TableauModel:
    """src/tableau.ts:26-31"""
    tableau: Tableau
    sign: float
    variables: list  # [(key, coefficients)]
    integers: list   # 1-based column indices of integer (incl. binary) variables


def _is_array_index(key):
    """JS 'integer index' property names: canonical decimal strings 0 .. 2^32-2."""
    return isinstance(key, str) and key.isdigit() and (key == "0" or key[0] != "0") and int(key) < 4294967295


def entries(seq):
    """convertToIterable (src/tableau.ts:33-38).  A dict plays the role of a JS plain object, so
    its entries come in JS property order (Object.entries): integer-like keys ascending first, then
    the others in insertion order.  Anything else is taken as an iterable of (key, value) pairs."""
    if isinstance(seq, dict):
        idx = sorted((k for k in seq if _is_array_index(k)), key=int)
        if idx:
            rest = [k for k in seq if not _is_array_index(k)]
            return [(k, seq[k]) for k in idx + rest]
        return list(seq.items())
    return list(seq)


def _to_set(s):
    """convertToSet (src/tableau.ts:41-45): True | set"""
    if s is True:
        return True
    if s is False or s is None:
        return set()
    return s if isinstance(s, (set, frozenset)) else set(s)


def _get(constraint, name):
    v = constraint.get(name) if isinstance(constraint, dict) else getattr(constraint, name, None)
    return None if v is None else float(v)


class _Cells(dict):
    """Stand-in for the zeroed Float64Array while the model is walked: flat index -> last value written."""

    def arrays(self, width):
        idx = np.fromiter(self.keys(), np.int64, len(self))
        val = np.fromiter(self.values(), np.float64, len(self))
        order = np.argsort(idx, kind="stable")
        idx, val = idx[order], val[order]
        return (idx // width).astype(np.int32), (idx % width).astype(np.int32), val


def tableau_model(model, sparse=False):
    """src/tableau.ts:47-137, line for line in behaviour.  sparse=True keeps only the cells the
    reference writes into its zeroed matrix (Tableau.cells; for yalps_tableau_assemble)."""
    sign = -1.0 if model.get("direction") == "minimize" else 1.0  # :51
    objective = model.get("objective")
    integers, binaries = model.get("integers"), model.get("binaries")
    constraints_iter = entries(model.get("constraints", {}))
    variables = entries(model.get("variables", {}))

    binary_constraint_col, ints = [], []  # :57-71
    if integers is not None or binaries is not None:
        binary_vars = _to_set(binaries)
        integer_vars = True if binary_vars is True else _to_set(integers)
        for i in range(1, len(variables) + 1):
            key = variables[i - 1][0]
            if binary_vars is True or key in binary_vars:
                binary_constraint_col.append(i)
                ints.append(i)
            elif integer_vars is True or key in integer_vars:
                ints.append(i)

    # :73-80 merge same-key constraints to the tightest [lower, upper]; `equal` beats min/max
    bounds = {}
    for key, constraint in constraints_iter:
        b = bounds.get(key)
        fresh = b is None
        if fresh:
            b = {"row": -1, "lower": -math.inf, "upper": math.inf}
        eq, mn, mx = _get(constraint, "equal"), _get(constraint, "min"), _get(constraint, "max")
        b["lower"] = max(b["lower"], eq if eq is not None else (mn if mn is not None else -math.inf))
        b["upper"] = min(b["upper"], eq if eq is not None else (mx if mx is not None else math.inf))
        if fresh:
            bounds[key] = b

    num_constraints = 1  # :82-86 rows in first-seen key order, upper row first then lower row
    for b in bounds.values():
        b["row"] = num_constraints
        num_constraints += (1 if math.isfinite(b["lower"]) else 0) + (1 if math.isfinite(b["upper"]) else 0)
    width = len(variables) + 1
    height = num_constraints + len(binary_constraint_col)
    num_vars = width + height
    m = _Cells() if sparse else np.zeros(width * height, np.float64)
    pos = np.arange(num_vars, dtype=np.int32)  # :95-98 identity
    var = np.arange(num_vars, dtype=np.int32)

    for c in range(1, width):  # :100-118 (a later duplicate coefficient overwrites an earlier one)
        for constraint, coef in entries(variables[c - 1][1]):
            coef = float(coef)
            if constraint == objective and objective is not None:
                m[c] = sign * coef
            b = bounds.get(constraint)
            if b is not None:
                if math.isfinite(b["upper"]):
                    m[b["row"] * width + c] = coef
                    if math.isfinite(b["lower"]):
                        m[(b["row"] + 1) * width + c] = -coef
                elif math.isfinite(b["lower"]):
                    m[b["row"] * width + c] = -coef

    for b in bounds.values():  # :120-128 RHS column
        if math.isfinite(b["upper"]):
            m[b["row"] * width] = b["upper"]
            if math.isfinite(b["lower"]):
                m[(b["row"] + 1) * width] = -b["lower"]
        elif math.isfinite(b["lower"]):
            m[b["row"] * width] = -b["lower"]

    for i, col in enumerate(binary_constraint_col):  # :130-134 x <= 1 rows for binaries
        row = num_constraints + i
        m[row * width] = 1.0
        m[row * width + col] = 1.0

    if sparse:
        return TableauModel(Tableau(None, width, height, pos, var, cells=m.arrays(width)), sign, variables, ints)
    return TableauModel(Tableau(m, width, height, pos, var), sign, variables, ints)


# src/constraint.ts:7-25
def less_eq(value):
    return {"max": value}


def greater_eq(value):
    return {"min": value}


def equal_to(value):
    return {"equal": value}


def in_range(lower, upper):
    return {"min": lower, "max": upper}
