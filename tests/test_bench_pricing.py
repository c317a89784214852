"""The benchmark's VALU pricing: instruction counters by class at their issue costs, the rest at the mean cost of such
instructions in the shipped kernels' ISA (profiles/valu_issue_mix.json).  No GPU needed."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_instruction_mix_file_covers_the_kernels_of_the_headline_frame():
    ks = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_mix.json")))["kernels"]
    for name in ("k_trace<false, false>", "k_trace<false, true>", "k_shadow<false, true, false>", "k_surface<true, false, false>"):
        k = ks[name]
        assert k["valu"] > 100
        assert 2.0 <= k["rest_mean_cost"] <= 4.5 and 2.0 <= k["mean_cost"] <= 4.5
        assert k["cycles"] == sum(c["cycles"] for c in k["classes"].values())


def test_issue_cycles_prices_every_class():
    rest, mean = bench.static_mix("k_trace")
    assert 2.0 < rest <= 4.5 and 2.0 < mean <= 4.5
    c = {"SQ_INSTS_VALU": 1000.0, "SQ_INSTS_VALU_FMA_F32": 100.0, "SQ_INSTS_VALU_MUL_F32": 50.0, "SQ_INSTS_VALU_ADD_F32": 50.0,
         "SQ_INSTS_VALU_CVT": 100.0, "SQ_INSTS_VALU_TRANS_F32": 10.0}
    cyc, how = bench.issue_cycles(c, "k_trace")
    assert abs(cyc - (200 * 2 + 100 * 4 + 10 * 8 + 690 * rest)) < 1e-6
    assert abs(how["rest_per_inst"] - 0.69) < 1e-9
    # without the class counters: every instruction at the kernel's mean cost
    cyc2, how2 = bench.issue_cycles({"SQ_INSTS_VALU": 1000.0}, "k_trace")
    assert abs(cyc2 - 1000 * mean) < 1e-6 and "mean_cost_static" in how2
    assert bench.issue_cycles({}, "k_trace") == (None, None)
    # an unknown kernel falls back to 4 cycles per instruction
    assert bench.static_mix("k_nothing") == (4.0, 4.0)
    # the fraction can never exceed what 8 cycles per instruction would give
    assert cyc <= 8 * 1000
