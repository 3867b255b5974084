"""tools/radius_load_model.py (profiles/r06_c_radius.md): the load model of k_radius counts what the document says for seed 0 -- the staged
formulation of tools/experiments/radius_staged.patch pulls at least 2.5 x fewer record bytes than one wavefront per centre, the ideal
grouping by edge bin at least 4 x, and the model's centre candidates are the oracle's own (asserted inside model())."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_staged_neighbourhoods_pull_a_third_of_the_record_bytes():
    import radius_load_model as rl
    from img2sgf_amd import synth
    from oracle import pipeline as opipe
    b = opipe.process_image(synth.synth_diagram(0)[0])["blurs"]
    tot = {}
    for plane in (b[5], b[7]):                                  # two of the four inputs that yield nearly all centres (Gaussian 3 x 3, 5 x 5)
        r = rl.model(plane)
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
    assert tot["centres"] > 1000
    assert tot["today_slots"] >= 2.5 * tot["patch_records"]
    assert tot["today_slots"] >= 4.0 * tot["staged_records"]
    assert tot["lds_records_max"] <= 2 * 1280                  # RAD_RCAP of the patch: the benchmark's neighbourhoods fit (model() takes max per plane, summed here)
