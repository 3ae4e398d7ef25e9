"""Fixed-column MPS reader + the netlib benchmark selection: host-side input plumbing for
BASELINE config 3.  Restates /root/reference/benchmarks/mps.ts (fields at
fixed columns :31-36, ROWS :70-100, COLUMNS :115-160, RHS :162-205, RANGES :207-245, BOUNDS
:247-302) and /root/reference/benchmarks/netlib/read.ts:18-58.  Not part of the accelerated path.
"""
import gzip
import json
import math
import os


class MPSError(ValueError):
    pass


def _f1(line):
    return line[1:3].strip()


def _f2(line):
    return line[4:12].strip()


def _f3(line):
    return line[14:22].strip()


def _f4(line):
    return line[24:36].strip()


def _f5(line):
    return line[39:47].strip()


def _f6(line):
    return line[49:61].strip()


def _num(text, what):
    if text == "":
        raise MPSError(f"Missing {what}")
    try:
        return float(text)
    except ValueError:
        raise MPSError(f"Failed to parse number '{text}'") from None


def model_from_mps(text, direction=None):
    """mps.ts:304-325 `modelFromMps`.  Returns a dict: name, direction, objective,
    constraints {row: [lower, upper]}, variables {col: {row: coef}}, bounds, integers, binaries.
    OBJSENSE / OBJNAME / SOS are not supported (like the reference); '*' lines are comments."""
    lines = text.replace("\r\n", "\n").split("\n")
    m = {"name": "", "direction": direction, "objective": None, "constraints": {}, "variables": {},
         "bounds": {}, "integers": set(), "binaries": set()}
    types = {}
    i = next((k for k, ln in enumerate(lines) if ln.startswith("NAME")), -1)
    if i < 0:
        raise MPSError("No NAME section was found")
    m["name"] = _f3(lines[i])
    idx = i + 1

    def next_line():
        nonlocal idx
        for k in range(idx + 1, len(lines)):
            if not lines[k].startswith("*"):
                idx = k
                return lines[k]
        return None

    def in_section(line):
        return line is not None and line.startswith(" ")

    def section():
        return lines[idx].rstrip() if idx < len(lines) else None

    def expect(name):
        if section() != name:
            raise MPSError(f"Line {idx + 1}: Expected section {name} but got {section()!r}")

    # ROWS
    expect("ROWS")
    line = next_line()
    while in_section(line):
        name, typ = _f2(line), _f1(line)
        if name == "":
            raise MPSError("Missing row name")
        if name in types:
            raise MPSError(f"The row '{name}' was already defined")
        if typ == "L":
            m["constraints"][name] = [-math.inf, 0.0]
        elif typ == "G":
            m["constraints"][name] = [0.0, math.inf]
        elif typ == "E":
            m["constraints"][name] = [0.0, 0.0]
        elif typ == "N":
            if m["objective"] is None:
                m["objective"] = name
            m["constraints"][name] = [-math.inf, math.inf]
        else:
            raise MPSError(f"Unexpected row type '{typ}'")
        types[name] = typ
        line = next_line()

    # COLUMNS
    expect("COLUMNS")

    def add_coef(variable, row, value):
        if row == "":
            raise MPSError("Missing row name")
        if row not in types:
            raise MPSError(f"The row '{row}' was not defined in the ROWS section")
        if row in variable:
            raise MPSError(f"The coefficient for row '{row}' was previously set for this column")
        variable[row] = _num(value, "coefficient value")

    integer_marked = False
    line = next_line()
    while in_section(line):
        if _f3(line) == "'MARKER'":
            marker = _f4(line)
            if marker == "'INTORG'":
                integer_marked = True
            elif marker == "'INTEND'":
                integer_marked = False
            else:
                raise MPSError(f"Unexpected MARKER '{marker}'")
            line = next_line()
            continue
        name = _f2(line)
        if name == "":
            raise MPSError("Missing column name")
        if name in m["variables"]:
            raise MPSError(f"Values for the column '{name}' were previously provided")
        variable = {}
        while True:
            add_coef(variable, _f3(line), _f4(line))
            if _f5(line) != "" or _f6(line) != "":
                add_coef(variable, _f5(line), _f6(line))
            line = next_line()
            if not (in_section(line) and _f2(line) == name):
                break
        m["variables"][name] = variable
        if integer_marked:
            m["integers"].add(name)

    # RHS
    expect("RHS")

    def add_rhs(row, value):
        if row == "":
            raise MPSError("Missing row name")
        typ = types.get(row)
        if typ is None:
            raise MPSError(f"The row '{row}' was not defined in the ROWS section")
        val = _num(value, "rhs value")
        c = m["constraints"][row]
        if typ in ("L", "E"):
            c[1] = val
        if typ in ("G", "E"):
            c[0] = val

    line = next_line()
    while in_section(line):
        add_rhs(_f3(line), _f4(line))
        if _f5(line) != "" or _f6(line) != "":
            add_rhs(_f5(line), _f6(line))
        line = next_line()

    # RANGES (mps.ts:207-227)
    if section() == "RANGES":
        def add_range(row, value):
            if row == "":
                raise MPSError("Missing row name")
            typ = types.get(row)
            if typ is None:
                raise MPSError(f"The row '{row}' was not defined in the ROWS section")
            val = _num(value, "range value")
            b = m["constraints"][row]
            if typ == "L" or (typ == "E" and val < 0.0):
                b[0] = b[1] - abs(val)
            if typ == "G" or (typ == "E" and val > 0.0):
                b[1] = b[0] + abs(val)

        line = next_line()
        while in_section(line):
            add_range(_f3(line), _f4(line))
            if _f5(line) != "" or _f6(line) != "":
                add_range(_f5(line), _f6(line))
            line = next_line()

    # BOUNDS (mps.ts:247-302)
    if section() == "BOUNDS":
        def set_bounds(name, lower, upper):
            b = m["bounds"].setdefault(name, [0.0, math.inf])
            if not math.isnan(lower):
                b[0] = lower
            if not math.isnan(upper):
                b[1] = upper

        line = next_line()
        while in_section(line):
            typ, col = _f1(line), _f3(line)
            if col == "":
                raise MPSError("Missing column name")
            if col not in m["variables"]:
                raise MPSError(f"The column '{col}' was not defined in the COLUMNS section")
            val = _num(_f4(line), "bound value") if typ in ("LO", "UP", "FX", "LI", "UI") else math.nan
            if typ == "LO":
                set_bounds(col, val, math.inf)
            elif typ == "UP":
                set_bounds(col, 0.0, val)
            elif typ == "FX":
                set_bounds(col, val, val)
            elif typ == "FR":
                set_bounds(col, -math.inf, math.inf)
            elif typ == "MI":
                set_bounds(col, -math.inf, 0.0)
            elif typ == "PL":
                set_bounds(col, 0.0, math.inf)
            elif typ == "BV":
                m["binaries"].add(col)
            elif typ == "LI":
                m["integers"].add(col)
                set_bounds(col, val, math.inf)
            elif typ == "UI":
                m["integers"].add(col)
                set_bounds(col, 0.0, val)
            else:
                raise MPSError(f"Unexpected bound type '{typ}'")
            line = next_line()
    expect("ENDATA")
    return m


def convert_constraints(bounds_map):
    """netlib/read.ts:18-30: [lower, upper] pairs -> Constraint objects (free rows drop out)."""
    out = {}
    for key, (lo, hi) in bounds_map.items():
        if math.isfinite(lo) and math.isfinite(hi):
            out[key] = {"equal": lo} if lo == hi else {"min": lo, "max": hi}
        elif math.isfinite(lo):
            out[key] = {"min": lo}
        elif math.isfinite(hi):
            out[key] = {"max": hi}
    return out


# netlib/read.ts:55-58: problems the reference itself cannot handle
TIMEOUT = ("25FV47", "AGG", "BANDM", "BNL1", "BRANDY", "DEGEN2", "DEGEN3", "E226", "FFFFF800", "SCFXM2", "SCFXM3",
           "SCSD1", "SCSD8", "STOCFOR2", "WOOD1P", "KLEIN3")


def read_benchmarks(directory, names=None):
    """netlib/read.ts:32-53 `readBenchmarks`: the index entries that pass the reference's filters
    and whose .mps[.gz] file is present, as dicts {name, expected, model, options}."""
    from .solve import default_options
    with open(os.path.join(directory, "index.json")) as f:
        index = json.load(f)
    out = []
    for b in index:
        if b["name"] in TIMEOUT or (names is not None and b["name"] not in names):
            continue
        if not (10_000 <= b["rows"] * b["cols"] <= 6_400_000):
            continue
        base = os.path.join(directory, b["name"].lower() + ".mps")
        if os.path.exists(base):
            with open(base) as f:
                text = f.read()
        elif os.path.exists(base + ".gz"):
            with gzip.open(base + ".gz", "rt") as f:
                text = f.read()
        else:
            continue  # read.ts:44-47: unreadable files are skipped
        mps = model_from_mps(text, "minimize")
        if mps["bounds"]:
            continue  # read.ts:50: BOUNDS sections are unsupported
        model = {"direction": mps["direction"], "objective": mps["objective"],
                 "constraints": list(convert_constraints(mps["constraints"]).items()),
                 "variables": [(k, list(v.items())) for k, v in mps["variables"].items()],
                 "integers": mps["integers"], "binaries": mps["binaries"]}
        options = dict(default_options)
        options.update(b.get("options") or {})
        out.append({"name": mps["name"], "expected": b["value"] if b["value"] is not None else math.nan,
                    "model": model, "options": options})
    return out
