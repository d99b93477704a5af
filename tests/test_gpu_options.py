"""The option surface: svh_context_set_option takes the five product options and refuses the tests' A/B switches,
which live behind svh_test_set_option (include/stevi_hip_test.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import _capi  # noqa: E402

PUBLIC = {"census_float_overflow": (0, 1), "census_winner_shortcut": (0, 1), "census_sweep": (0, 1, 3), "sgm_score_fused": (0, 1, 2), "literal_cost_volumes": (0, 1)}
TEST_ONLY = ["census_fast_path", "census_sweep_rl", "census_tiles", "cost_volume_colsum", "patchmatch_pred_costs", "patchmatch_run_batches", "patchmatch_lookback", "patchmatch_scan_chunks", "patchmatch_search_form", "feature_volume_tiled", "feature_volume_records", "extract_index_wide", "guided_shared",
             "sgm_score_pad", "fold_2d_offsets", "cost_reduce_fused", "sgm_cost_two_minima", "sgm_score_finish_fused"]
DEFAULTS = {"census_float_overflow": 0, "census_winner_shortcut": 1, "census_sweep": 0, "sgm_score_fused": 1, "literal_cost_volumes": 0}


def test_public_options_and_their_values():
    t = torch.zeros(1, device="cuda:0")
    for name, values in PUBLIC.items():
        for v in values:
            sv.set_option(t, name, v)
        sv.set_option(t, name, DEFAULTS[name])
    for name, bad in (("census_float_overflow", 2), ("census_sweep", 2), ("sgm_score_fused", 3), ("sgm_score_fused", -1)):
        with pytest.raises(_capi.SvhError) as e:
            sv.set_option(t, name, bad)
        assert e.value.status == _capi.ERR_INVALID_ARGUMENT


def test_test_switches_are_not_on_the_public_entry_point():
    t = torch.zeros(1, device="cuda:0")
    for name in TEST_ONLY + ["no_such_option"]:
        with pytest.raises(_capi.SvhError) as e:
            sv.set_option(t, name, 1)
        assert e.value.status == _capi.ERR_INVALID_ARGUMENT and "unknown option" in str(e.value)
    for name in TEST_ONLY:
        sv.set_test_option(t, name, 0)
        sv.set_test_option(t, name, 1)
    sv.set_test_option(t, "sgm_score_fused", 3)
    sv.set_test_option(t, "sgm_score_fused", 1)
    sv.set_test_option(t, "guided_shared", 3)
    sv.set_test_option(t, "guided_shared", 1)
    for name, v in DEFAULTS.items():  # the test entry point drives the public options too
        sv.set_test_option(t, name, v)
    with pytest.raises(_capi.SvhError):
        sv.set_test_option(t, "no_such_option", 1)
