"""Scenario I/O in the RCP emissions CSV layout."""
import os

import numpy as np
import pytest

from fiveeqscm_amd import emissions as emi
from fiveeqscm_amd.scenario import read_emissions_csv, write_emissions_csv

HERE = os.path.dirname(os.path.abspath(__file__))


def test_read_rcp_layout_fixture():
    years, E = read_emissions_csv(os.path.join(HERE, "golden", "rcp_layout_sample.csv"))
    assert years.tolist() == [1765, 1766, 1767, 1768, 1769]
    np.testing.assert_allclose(E[:, 0], [0.253, 0.264, 0.275, 0.286, 0.298])      # Fossil + Other
    assert E[:, 1].tolist() == [10.5, 10.9, 11.3, 11.7, 12.1] and E[:, 2].tolist() == [0.61, 0.62, 0.63, 0.64, 0.65]
    _, co2 = read_emissions_csv(os.path.join(HERE, "golden", "rcp_layout_sample.csv"), gases=("CO2",))
    assert co2.shape == (5, 1)
    with pytest.raises(ValueError, match="no column for HFC23"):
        read_emissions_csv(os.path.join(HERE, "golden", "rcp_layout_sample.csv"), gases=("CO2", "HFC23"))


def test_write_read_roundtrip_is_bit_exact(tmp_path):
    E = emi.rcp_like_emissions(750, 3)
    years = 1765 + np.arange(750)
    path = tmp_path / "scen.csv"
    write_emissions_csv(path, years, E)
    y2, E2 = read_emissions_csv(path)
    assert np.array_equal(y2, years) and E2.tobytes() == E.tobytes()
    assert emi.emissions_sha256(E2) == emi.emissions_sha256(E)
    np.testing.assert_array_equal(emi.make_drive(E2), emi.make_drive(E))


def test_malformed_files(tmp_path):
    p = tmp_path / "a.csv"
    p.write_text("just text\nmore text\n")
    with pytest.raises(ValueError, match="no column-name row"):
        read_emissions_csv(p)
    p.write_text("YEARS,FossilCO2,CH4,N2O\n")
    with pytest.raises(ValueError, match="no numeric data"):
        read_emissions_csv(p)
    p.write_text("YEARS,FossilCO2,CH4,N2O\n2000,1,2,3\n2002,1,2,3\n2003,1,2,3\n")
    with pytest.raises(ValueError, match="equal steps"):
        read_emissions_csv(p)
    p.write_text("YEARS,FossilCO2,CH4,N2O\n2000,1,,3\n2001,1,2,3\n")
    with pytest.raises(ValueError, match="missing values"):
        read_emissions_csv(p)
    with pytest.raises(ValueError):
        write_emissions_csv(p, [2000, 2001], np.zeros((3, 3)))


@pytest.mark.gpu
def test_a_csv_scenario_through_the_engine(tmp_path):
    """SURVEY section 8f-4 on the GPU: (i) the frozen 750-step scenario written in the RCP layout, read back and run from the
    FILE gives the bits of the run from the ARRAY (C, T, state), 20k members x 3 gases; (ii) the RCP-layout fixture
    tests/golden/rcp_layout_sample.csv (FossilCO2 + OtherCO2 summed, extra species ignored) through a 5-step run against the
    oracle at <= 1e-10 relative — per-step and fused launches."""
    import torch

    from fiveeqscm_amd import params as prm
    from fiveeqscm_amd.engine import EnsembleEngine
    from oracle import fiveeq_oracle as npo
    N = 20_000
    p = prm.sample_ensemble_shard(prm.default_params("multigas"), N, device="cuda:0")
    E = emi.rcp_like_emissions(750, 3)
    path = tmp_path / "rcp_like.csv"
    write_emissions_csv(path, 1765 + np.arange(750), E)
    years, E_file = read_emissions_csv(path)
    assert years[0] == 1765 and years[-1] == 2514
    stored = sorted(set(list(range(0, 750, 50)) + [249, 499, 749]))
    a = EnsembleEngine(p, N, E, device="cuda:0", output_steps=stored)
    b = EnsembleEngine(p, N, E_file, device="cuda:0", output_steps=stored)
    a.run(mode="per_step")
    b.run(mode="fused")
    torch.cuda.synchronize()
    for name in ("C", "T", "R", "S"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.drive, b.drive)
    # ... and out again: the end-of-run summary of the run as a CSV, bit for bit
    from fiveeqscm_amd.scenario import read_summary_csv, write_summary_csv
    summ = b.gather_summary([249, 499, 749])
    write_summary_csv(tmp_path / "summary.csv", years[[249, 499, 749]], {k: (v.numpy() if v is not None else None) for k, v in summ.items()})
    y3, back = read_summary_csv(tmp_path / "summary.csv")
    assert y3.tolist() == [2014.0, 2264.0, 2514.0] and np.array_equal(back["percentiles"], summ["percentiles"].numpy())
    rows = [stored.index(t) for t in (249, 499, 749)]
    assert np.array_equal(back["percentiles"], np.percentile(a.T[rows].cpu().numpy(), (5.0, 50.0, 95.0), axis=1).T)
    a.close(), b.close()
    _, E5 = read_emissions_csv(os.path.join(HERE, "golden", "rcp_layout_sample.csv"))
    n = 777
    ph = prm.sample_ensemble(prm.default_params("multigas"), n)
    want = npo.run(E5, ph, n)
    for mode in ("per_step", "fused"):
        eng = EnsembleEngine(ph, n, E5, device="cuda:0")
        eng.run(mode=mode)
        torch.cuda.synchronize()
        for name in ("C", "T"):
            got = getattr(eng, name).cpu().numpy()
            assert np.all(np.abs(got - want[name]) <= 1e-10 * np.abs(want[name]) + 1e-13), (mode, name)
        eng.close()


def test_summary_csv_roundtrip(tmp_path):
    """The output side of scenario I/O: an end-of-run summary (moments + exact percentiles per output year) written as CSV and
    read back bit for bit; a summary without percentiles (a non-root rank's) is refused."""
    from fiveeqscm_amd.scenario import read_summary_csv, write_summary_csv
    rng = np.random.default_rng(4)
    K = 3
    summ = {"count": np.full(K, 1e6), "mean": rng.normal(2, 1, K), "var": rng.uniform(0.1, 1, K), "min": rng.normal(0, 1, K),
            "max": rng.normal(5, 1, K), "percentiles": np.sort(rng.normal(2, 1, (K, 4)), axis=1)}
    path = tmp_path / "summary.csv"
    write_summary_csv(path, [2014, 2264, 2514], summ, percentiles=(5, 50, 95, 99.5))
    years, back = read_summary_csv(path)
    assert years.tolist() == [2014.0, 2264.0, 2514.0] and back["levels"] == [5.0, 50.0, 95.0, 99.5]
    assert np.array_equal(back["percentiles"], summ["percentiles"]) and np.array_equal(back["mean"], summ["mean"])
    assert np.array_equal(back["std"], np.sqrt(summ["var"])) and np.array_equal(back["min"], summ["min"]) and back["count"].tolist() == [1e6] * 3
    with pytest.raises(ValueError, match="root rank"):
        write_summary_csv(path, [1, 2, 3], dict(summ, percentiles=None))
    with pytest.raises(ValueError, match="does not match"):
        write_summary_csv(path, [1, 2], summ, percentiles=(5, 50, 95, 99.5))
