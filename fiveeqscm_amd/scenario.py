"""Scenario I/O: emissions time series in the RCP / MAGICC "EMISSIONS.csv" column layout.

SURVEY.md section 8f-4 ("scenario I/O (CSV emissions in RCP column layout)").  The reference
(stujen/fiveEqSCM @ v0) ships no data files and no reader, so the layout
followed here is the public one of the RCP emissions files that FaIR-family models read:
free-text header lines, then a row of column names whose first cell holds the word YEARS and
whose other cells name the species (FossilCO2, OtherCO2, CH4, N2O, ...), optionally a UNITS row,
then one row per year: year, value, value, ...   Units: CO2 GtC/yr (FossilCO2 + OtherCO2 are
summed), CH4 Mt CH4/yr, N2O Mt N2O-N/yr — the units of the engine's default parameter sets.
"""
import csv

import numpy as np

_CO2_PARTS = ("FossilCO2", "OtherCO2")
_ALIASES = {"CO2": ("CO2", "CO2_total", "TotalCO2"), "CH4": ("CH4",), "N2O": ("N2O",)}


def _is_number(cell):
    try:
        float(cell)
        return True
    except ValueError:
        return False


def read_emissions_csv(path, gases=("CO2", "CH4", "N2O")):
    """Return (years [n] float64, emissions [n, len(gases)] float64) from an RCP-layout CSV.

    The column-name row is the last non-numeric row before the data whose cells include every
    requested species (CO2 may be given as FossilCO2 [+ OtherCO2] or as one CO2 column).  Rows
    must be in increasing, evenly spaced years.  Raises ValueError on a malformed file."""
    with open(path, newline="") as fh:
        rows = [[c.strip() for c in row] for row in csv.reader(fh)]
    header, data = None, []
    for row in rows:
        if not row or all(c == "" for c in row):
            continue
        if _is_number(row[0]) and len(row) > 1 and all(_is_number(c) or c == "" for c in row[1:]):
            data.append(row)
        elif not data:
            names = [c.replace(" ", "") for c in row]
            if any(n in names for n in _CO2_PARTS + _ALIASES["CO2"] + ("CH4", "N2O")):
                header = names
    if header is None:
        raise ValueError(f"{path}: no column-name row found (expected species names such as FossilCO2, CH4, N2O)")
    if not data:
        raise ValueError(f"{path}: no numeric data rows")
    width = len(header)
    table = np.array([[float(c) if c != "" else np.nan for c in (row + [""] * width)[:width]] for row in data])
    years = table[:, 0]
    if years.size > 1:
        step = np.diff(years)
        if np.any(step <= 0) or not np.allclose(step, step[0]):
            raise ValueError(f"{path}: years must increase in equal steps")

    def col(name):
        return table[:, header.index(name)]

    cols = []
    for gas in gases:
        if gas == "CO2" and _CO2_PARTS[0] in header:
            v = col(_CO2_PARTS[0]) + (col(_CO2_PARTS[1]) if _CO2_PARTS[1] in header else 0.0)
        else:
            name = next((n for n in _ALIASES.get(gas, (gas,)) if n in header), None)
            if name is None:
                raise ValueError(f"{path}: no column for {gas} (have {header[1:]})")
            v = col(name)
        if np.any(~np.isfinite(v)):
            raise ValueError(f"{path}: missing values in the {gas} column")
        cols.append(v)
    return years, np.stack(cols, axis=1)


def write_emissions_csv(path, years, emissions, gases=("CO2", "CH4", "N2O"), title="fiveeqscm_amd scenario"):
    """Write [n, len(gases)] emissions in the same layout (CO2 goes to FossilCO2, OtherCO2 = 0),
    with repr-exact floats so that read(write(x)) == x bit for bit."""
    E = np.asarray(emissions, dtype=np.float64)
    years = np.asarray(years, dtype=np.float64)
    if E.ndim != 2 or E.shape != (years.size, len(gases)):
        raise ValueError(f"emissions shape {E.shape} does not match {years.size} years x {len(gases)} gases")
    units = {"CO2": "GtC", "CH4": "MtCH4", "N2O": "MtN2O-N"}
    names, unit_row = [], []
    for g in gases:
        if g == "CO2":
            names += list(_CO2_PARTS)
            unit_row += ["GtC", "GtC"]
        else:
            names.append(g)
            unit_row.append(units.get(g, ""))
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow([title])
        w.writerow(["COLUMNS:", len(names)])
        w.writerow(["UNITS"] + unit_row)
        w.writerow(["v YEARS/GAS >"] + names)
        for y, row in zip(years, E):
            cells = []
            for g, v in zip(gases, row):
                cells += [repr(float(v)), "0.0"] if g == "CO2" else [repr(float(v))]
            w.writerow([repr(float(y)) if y != int(y) else int(y)] + cells)


def write_summary_csv(path, years, summary, percentiles=(5.0, 50.0, 95.0), quantity="T", unit="K"):
    """The output side: the end-of-run summary of an ensemble (the dict `distributed.gather_summary` /
    `EnsembleEngine.gather_summary` return on the root rank: count, mean, var, min, max [K] and percentiles [K, P]) as one CSV
    row per output year — YEAR, COUNT, MEAN, STD, MIN, P05, P50, P95, MAX (columns named after `percentiles`) — with repr-exact
    floats, so that `read_summary_csv` returns the numbers bit for bit."""
    years = np.asarray(years, dtype=np.float64).reshape(-1)
    cols = {k: np.asarray(summary[k], dtype=np.float64).reshape(-1) for k in ("count", "mean", "var", "min", "max")}
    if summary.get("percentiles") is None:
        raise ValueError("summary holds no percentiles (they exist on the root rank of the exchange only)")
    pct = np.asarray(summary["percentiles"], dtype=np.float64).reshape(years.size, -1)
    if pct.shape[1] != len(percentiles) or any(v.size != years.size for v in cols.values()):
        raise ValueError(f"summary of {cols['mean'].size} rows x {pct.shape[1]} percentiles does not match {years.size} years x "
                         f"{len(percentiles)} percentiles")
    names = ["P%s" % (("%02d" % p) if float(p).is_integer() else repr(float(p))) for p in percentiles]
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow([f"fiveeqscm_amd ensemble summary of {quantity} ({unit})"])
        w.writerow(["YEAR", "COUNT", "MEAN", "STD", "MIN"] + names + ["MAX"])
        for k, y in enumerate(years):
            w.writerow([int(y) if y == int(y) else repr(float(y)), int(cols["count"][k]), repr(float(cols["mean"][k])),
                        repr(float(np.sqrt(cols["var"][k]))), repr(float(cols["min"][k]))]
                       + [repr(float(v)) for v in pct[k]] + [repr(float(cols["max"][k]))])


def read_summary_csv(path):
    """(years [K], dict of columns) of a file written by `write_summary_csv`; the percentile columns come back as
    'percentiles' [K, P] with their levels in 'levels'."""
    with open(path, newline="") as fh:
        rows = [row for row in csv.reader(fh) if row]
    header = next((r for r in rows if r and r[0] == "YEAR"), None)
    if header is None:
        raise ValueError(f"{path}: no YEAR header row")
    data = np.array([[float(c) for c in r] for r in rows[rows.index(header) + 1:]], dtype=np.float64).reshape(-1, len(header))
    out = {name.lower(): data[:, i] for i, name in enumerate(header) if not name.startswith("P") and name != "YEAR"}
    pcols = [i for i, name in enumerate(header) if name.startswith("P")]
    out["levels"] = [float(header[i][1:]) for i in pcols]
    out["percentiles"] = data[:, pcols]
    return data[:, 0], out
