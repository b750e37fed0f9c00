"""CPU checks of small pieces the measurements and the multi-GPU defaults rest on (no GPU, no oracle)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_history_halo_default_scales_with_the_frame_height():
    from raytracingdenoiser_amd import sharding

    rows = [sharding.HaloSharder.default_motion_rows(h) for h in (128, 720, 1080, 1440, 2160, 4320)]
    assert rows == [32, 32, 32, 32, 48, 96]  # 32 up to 1440p (the value every committed measurement used), proportional above
    assert rows == sorted(rows)


def test_counter_summaries_fold_into_valu_figures():
    """tools/pmc_to_json.py on the committed counter summaries: the per-kernel VALU figures bench.py's roofline.valu is built from"""
    import pmc_to_json

    sq = pmc_to_json.parse_columns(os.path.join(ROOT, "profiles", "r03_i_reblur_ds_pmc3.txt"))
    ta = [v for k, v in sq.items() if "ReblurTemporalAccumulationKernel<true, true, false, 0, false, 3, 1>" in k]
    assert len(ta) == 1 and ta[0]["SQ_WAVES"] == 57600.0  # 2560 x 1440 / 64
    per_wave = ta[0]["SQ_INSTS_VALU"] / ta[0]["SQ_WAVES"]
    assert 2000 < per_wave < 3000
    cycles_per_instruction = 4.0 * ta[0]["SQ_ACTIVE_INST_V"] / ta[0]["SQ_INSTS_VALU"]  # the counter's unit is 4 cycles
    assert 3.9 < cycles_per_instruction < 4.4
    fetch = pmc_to_json.parse(os.path.join(ROOT, "profiles", "r03_i_reblur_ds_pmc1.txt"))
    assert any("ReblurSpatialKernel" in k for k in fetch) and all(v >= 0 for v in fetch.values())


def test_lds_array_pricing_prefers_whole_texel_reads():
    """tools/isa_stats.py: the rule that found RELAX HistoryClamping's bound -- a float4 texel read as b96 + b32 costs 4x one b128"""
    import isa_stats

    split = collections.Counter({"ds_read_b96": 102, "ds_read_b32": 51})
    whole = collections.Counter({"ds_read_b128": 104})
    assert isa_stats.lds_cycles(whole) == 416
    assert isa_stats.lds_cycles(split) >= 2 * isa_stats.lds_cycles(whole)  # (conflict-free pricing; the b32 reads at a 16-byte stride conflict 4-way on top)
