"""ER / SCR agents with the reference's class names and method signatures (agents/*.py)."""
