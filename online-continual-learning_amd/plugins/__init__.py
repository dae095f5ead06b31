"""retrieve / update plugins with the reference's class names and call signatures (utils/buffer/*.py)."""
