"""tools/median_network.py (profiles/r06_b_median.md): the generated sorting-network medians are exact, and the counts the decision about
k_median57 rests on are what the document says."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import median_network as mn


def test_networks_are_exact_medians():
    assert mn.check(5, 2, "tile2")
    assert mn.check(7, 2, "tile2q", col7from5=True)
    assert mn.check(7, 4, "tile2")
    assert mn.check(5, 1, "independent")


def test_counts_of_the_best_schemes():
    assert mn.marginal(5, 2, "tile2") == 53.0
    assert mn.marginal(7, 4, "tile2q") == 116.5
    assert mn.marginal(7, 2, "tile2q", True, with5="tile2") == 167.0       # both medians in one graph, 7-columns from the 5-columns
