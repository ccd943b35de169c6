"""A seeded 20-second slice of tools/fuzz/differential.py under `pytest -m gpu`: random shapes and
inputs over the C ABI's entry points against the oracle -- sizes that are not powers of two, empty inputs, ragged batches,
special scalars, points at infinity.  The long runs (quarter of an hour, profiles/r02z_differential_fuzz.txt) stay a tool."""
import importlib.util, os
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_differential_fuzz_slice_has_no_difference():
    spec = importlib.util.spec_from_file_location("mzk_differential", os.path.join(HERE, "..", "tools", "fuzz", "differential.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    counts, failures = mod.run(budget=20.0, seed=20261002)
    assert not failures, failures[:5]
    assert sum(counts.values()) >= 200 and len(counts) >= 10, counts       # the slice reached most entry points
