"""Every sweep-kernel template instantiation some dispatch of libkrotov_hip.so can select must have been launched by
an oracle-comparing GPU test of this session (a published number must not come from code no test
checks).  The library keeps the registry itself (``kh_debug_launched``: naming a kernel in ``launch_plain<>`` /
``launch_persistent<>`` registers it), tests/conftest.py points ``KH_LAUNCH_LOG`` at one file for the whole session
while a GPU test without the ``no_oracle`` marker runs (rank sub-processes inherit it).  The file sorts last on
purpose; it only judges a full run (``pytest tests -m gpu`` with no ``-k`` and no single files)."""
import pytest


def test_registry_lists_the_dispatchable_instantiations():
    """Host only: the registry is filled when the library is loaded and names every family."""
    from krotov_amd import _lib

    names = _lib.kernel_instantiations()
    assert len(names) == len(set(names)) and len(names) > 100
    for family in ('kh_q2_forward_update<', 'kh_q2_sweep_store', 'kh_tile_forward_update<', 'kh_stream_forward_update<',
                   'kh_ens_forward_update<', 'kh_coop_forward_update<', 'kh_tn_forward_update<',
                   'kh_ell_forward_update<', 'kh_mini_forward_update<', 'kh_quad_forward_update<', 'kh_gen_forward_update',
                   'kh_gen_sweep_store'):
        assert any(n.startswith(family) for n in names), family


@pytest.mark.gpu
@pytest.mark.no_oracle
def test_every_dispatchable_instantiation_ran_under_an_oracle_test(request):
    from krotov_amd import _lib

    if not request.config._kh_full_run:
        pytest.skip("only a full `pytest tests -m gpu` run is judged")
    with open(request.config._kh_launch_log) as f:
        launched = {line.strip() for line in f if line.strip()}
    names = _lib.kernel_instantiations()
    missing = [n for n in names if n not in launched]
    print("%d of %d instantiations launched under oracle-comparing tests" % (len(names) - len(missing), len(names)))
    assert not missing, "never launched by an oracle-comparing test:\n  " + "\n  ".join(missing)
