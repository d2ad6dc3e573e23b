"""The searched cell genotype both entry scripts instantiate (test_original.py:709-713, robust_test.py:253-257)."""
import collections

Genotype = collections.namedtuple("Genotype", "normal_1 normal_1_concat normal_2 normal_2_concat normal_3 normal_3_concat")

fusion_at = Genotype(
    normal_1=[("Denseblocks_3_1", 0), ("DilConv_3_2", 1)], normal_1_concat=[1, 2],
    normal_2=[("Denseblocks_3_1", 0), ("Denseblocks_3_1", 1)], normal_2_concat=[1, 2],
    normal_3=[("ECAattention_3", 0), ("Residualblocks_7_1", 1)], normal_3_concat=[1, 2],
)
FUSION_AT = fusion_at
