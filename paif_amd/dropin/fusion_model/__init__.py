"""Competitor fusion networks of the paper's tables (reference fusion_model/*): out of scope.
test_original.py:18-19 imports DID and ReCoNet and never constructs them; the names resolve here and raise when built."""
