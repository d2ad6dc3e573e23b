"""Oracle shim: empty stand-in for cv2 (attack/attack.py:8, never called on the path)."""
