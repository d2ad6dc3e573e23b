"""`from core import ...` (reference core/__init__.py)."""
