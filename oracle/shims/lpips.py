"""Oracle shim: empty stand-in for lpips (attack/attack.py:6, never called on the path)."""
